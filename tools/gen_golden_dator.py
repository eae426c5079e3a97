#!/usr/bin/env python3
"""Golden vectors for DATOR: the reference's own `build_FourDNet`
(/root/reference/dator/model/make_model.py:424-843) and `TransReID` backbone (dator/model/backbones/vit_pytorch.py)
imported in the build container with local shims -- a stub for the absent, unused `cv2`, a SimpleNamespace in place of the
yacs config (PRETRAIN_CHOICE != 'imagenet' skips the checkpoint load), and integer CUDA ordinals redirected to the CPU --
and run on the seeded weights of ibloc_amd.dator.  Only outputs are stored (tests/golden/dator_golden.npz)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from ibloc_amd import dator as D  # noqa: E402

REF = "/root/reference/dator"
OUT = os.path.join(ROOT, "tests", "golden", "dator_golden.npz")
OUT_KEYS = os.path.join(ROOT, "tests", "golden", "dator_state_keys.json")


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def load_reference():
    for name in ("cv2",):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    # integer device ordinals -> cpu
    _mto, _tto = torch.nn.Module.to, torch.Tensor.to

    def mto(self, *a, **k):
        a = tuple("cpu" if isinstance(x, int) and not isinstance(x, bool) else x for x in a)
        return _mto(self, *a, **k)

    def tto(self, *a, **k):
        a = tuple("cpu" if isinstance(x, int) and not isinstance(x, bool) else x for x in a)
        return _tto(self, *a, **k)

    torch.nn.Module.to, torch.Tensor.to = mto, tto
    sys.path.insert(0, REF)
    import importlib
    mm = importlib.import_module("model.make_model")
    return mm


def stream_state(prefix, w, sd):
    sd[prefix + "cls_token"] = t(w["cls"]).reshape(1, 1, -1)
    sd[prefix + "pos_embed"] = t(w["pos"]).unsqueeze(0)
    sd[prefix + "patch_embed.proj.weight"] = t(w["patch.w"])
    sd[prefix + "patch_embed.proj.bias"] = t(w["patch.b"])
    for l in range(12):
        p, q = f"{prefix}blocks.{l}.", f"l{l}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = t(w[q + "ln1.g"]), t(w[q + "ln1.b"])
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = t(w[q + "ln2.g"]), t(w[q + "ln2.b"])
        sd[p + "attn.qkv.weight"] = torch.cat([t(w[q + "q.w"]), t(w[q + "k.w"]), t(w[q + "v.w"])], 0)
        sd[p + "attn.qkv.bias"] = torch.cat([t(w[q + "q.b"]), t(w[q + "k.b"]), t(w[q + "v.b"])], 0)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = t(w[q + "o.w"]), t(w[q + "o.b"])
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = t(w[q + "fc1.w"]), t(w[q + "fc1.b"])
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = t(w[q + "fc2.w"]), t(w[q + "fc2.b"])
        if q + "lora_down" in w:
            sd[p + "attn.qkv_lora_down_matrix"] = t(w[q + "lora_down"])
            sd[p + "attn.qkv_lora_up_matrix"] = t(w[q + "lora_up"])


def head_state(hw, sd):
    names = {"proj_local_rgb": "project_local_rgb", "proj_global_rgb": "project_global_rgb", "merge_rgb": "merge_local_global_rgb",
             "proj_local_depth": "project_local_depth", "proj_global_depth": "project_global_depth",
             "merge_depth": "merge_local_global_depth", "Q_r": "Q_r", "V_r": "V_r", "Q_d": "Q_d", "V_d": "V_d"}
    for mine, ref in names.items():
        sd[ref + ".weight"], sd[ref + ".bias"] = t(hw[mine + ".w"]), t(hw[mine + ".b"])
    for op in D.ATTN_OPS:
        sd[f"{op}_selector.0.weight"], sd[f"{op}_selector.0.bias"] = t(hw[op + ".sel.w"]), t(hw[op + ".sel.b"])
        sd[f"{op}_attn_weights.0.weight"], sd[f"{op}_attn_weights.0.bias"] = t(hw[op + ".aw.w"]), t(hw[op + ".aw.b"])
        sd[f"{op}_ffn.weight"], sd[f"{op}_ffn.bias"] = t(hw[op + ".ffn.w"]), t(hw[op + ".ffn.b"])
        sd[f"{op}_norm.weight"], sd[f"{op}_norm.bias"] = t(hw[op + ".norm.g"]), t(hw[op + ".norm.b"])
    for i in range(4):
        sd[f"hypernet.{2 * i}.weight"], sd[f"hypernet.{2 * i}.bias"] = t(hw[f"hyper.{i}.w"]), t(hw[f"hyper.{i}.b"])


def main():
    mm = load_reference()
    NS = types.SimpleNamespace
    cfg = NS(MODEL=NS(PRETRAIN_PATH="", PRETRAIN_CHOICE="none", NECK="bnneck", TRANSFORMER_TYPE="vit_base_patch16_224_TransReID",
                      SIE_CAMERA=False, SIE_VIEW=False, SIE_COE=3.0, JPM=True, STRIDE_SIZE=[16, 16], DROP_PATH=0.1),
             TEST=NS(NECK_FEAT="before"), INPUT=NS(SIZE_TRAIN=[256, 128]))
    torch.manual_seed(0)
    model = mm.build_FourDNet(10, 0, 0, cfg, mm.__factory_T_type, 0, 0, 0, rearrange=False)
    model.eval()
    rw, dw, hw = D.random_stream_weights(301), D.random_stream_weights(302), D.random_head_weights(303)
    sd = {}
    stream_state("base.", rw, sd)
    stream_state("base2.", dw, sd)
    head_state(hw, sd)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    allowed = ("classifier", "fc.", ".norm.", "sie_embed")
    missing = [k for k in missing if not any(a in k for a in allowed)]
    assert not unexpected and not missing, (missing[:10], unexpected[:10])
    rng = np.random.default_rng(304)
    rgb = rng.normal(size=(3, 3, 256, 128)).astype(np.float32)
    depth = np.repeat(rng.uniform(-1, 1, size=(3, 1, 256, 128)).astype(np.float32), 3, axis=1)
    with torch.no_grad():
        emb = model(torch.from_numpy(rgb), torch.from_numpy(depth)).numpy()
        rgb_tok = model.base(torch.from_numpy(rgb)).numpy()
    print("embedding", emb.shape, float(np.abs(emb).mean()), "rgb tokens", rgb_tok.shape)
    np.savez_compressed(OUT, embedding=emb.astype(np.float32), rgb_tokens_cls=rgb_tok[:, 0].astype(np.float32),
                        rgb_tokens_mean=rgb_tok.mean(1).astype(np.float32))
    print("wrote", OUT)
    # the checkpoint layout the reference's load_param reads (make_model.py:620-626): names and shapes of the model's own
    # state_dict() -- data for tests/test_converters.py -- and the round trip through the product's converter, checked here against
    # the live reference model: state_dict() (as a DataParallel checkpoint would carry it: `module.` prefix) -> weights == what was loaded
    import json
    ref_sd = model.state_dict()
    json.dump({k: list(v.shape) for k, v in ref_sd.items()}, open(OUT_KEYS, "w"), indent=0)
    print("wrote", OUT_KEYS, len(ref_sd), "entries")
    rw2, dw2, hw2 = D.fourdnet_state_dict_to_weights({"module." + k: v for k, v in ref_sd.items()})
    for got, want in ((rw2, rw), (dw2, dw), (hw2, hw)):
        assert set(want) - set(got) <= {"ln_f.g", "ln_f.b"} and set(got) <= set(want), set(got) ^ set(want)   # the streams run without their final norm
        for k in got:
            assert np.array_equal(got[k], np.asarray(want[k], dtype=np.float32).reshape(got[k].shape)), k
    print("converter round trip against the reference model's state_dict: identical")


if __name__ == "__main__":
    main()
