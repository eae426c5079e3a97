#!/usr/bin/env python3
"""bench.py -- query-frames localised / second (embed + match + register) on MI355X.

Default workload = the north-star row "T" (BASELINE.json metric): DINOv2 ViT-B/14 crops (224^2, Q = 7 per frame)
against a 10 000-instance memory (E = 4 stored embeddings each, 5 000-point coloured clouds), FPFH + RANSAC + coloured ICP on the
clouds of every candidate assignment, whole-memory evaluation, on one MI355X.  `--config` selects the other BASELINE configs:
    T   DINOv2-B/14, M = 10 000 (default; the row the >= 200 frames/s target is judged on)
    C2  DINOv2-B/14, M = 1 000          (BASELINE configs[1])
    C3  DATOR RGB-D dual stream (256x128 RGB + depth crops), M = 5 000, D = 128   (configs[2])
    C1  DINOv2-S/14 (D = 384), M = 20, Q = 4 crops of 64..400 px per frame (configs[0], the reference's CPU-runnable plumbing case)
    C5  one-GPU slice of configs[4]: 100 000-point objects, clouds sharded by instance range (--memory 6250 = one GPU's share of 50 000)
    C4  embed + match + assign only against M = 50 000 instances (configs[3]; one GPU holds the whole embedding memory, or its
        1/N instance range with --shard-memory under torch.distributed)
A "step" is one pass of the hot path (ObjectMemory.localise body, object_memory.py:911-1131) over one batch of --frames synthetic
query frames whose crops and detected clouds are already resident in HBM.  Consecutive steps are pipelined the way a localisation
service runs (LocaliseEngine.localise_stream: embed + match of step k + 1 on a second HIP stream while step k is assigned and
registered; --sequential runs them back to back); all work of the K timed steps happens inside the timed region.  Synthetic data and
seeded random-init weights (no datasets / checkpoints offline).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            # starts N ranks itself (child `python -m torch.distributed.run`, before any HIP call here)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU (one process per GPU, "scaling": "weak": every rank localises --frames frames per step).  Default layout at N > 1 is the
north star's: the embedding memory is sharded by instance range (`--layout sharded`), every rank matches all ranks' query rows against
its range and the per-shard two-ended top-k candidate lists are exchanged over the library's RCCL communicator before the assignment
search; clouds, resident features and the evaluation grid stay replicated while they fit 288 GB (--shard-clouds shards
them too).  `--layout replicated` is the communication-free alternative (frames are independent, tum_localisation_trial.py:215).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE configs[0]: the reference's own CPU-runnable case (TUM fr1/desk scale: DINOv2 ViT-S/14, 20-object memory, SURVEY 8d row C1:
    # Q = 4 crops per frame of sizes H, W ~ U{64..400} through the PIL-exact resize kernel, 5 000-point clouds)
    "C1": dict(model="dinov2_vits14", memory=20, points=5000, register=True, q=4, var_crops=True),
    "T": dict(model="dinov2_vitb14", memory=10000, points=5000, register=True),
    "C2": dict(model="dinov2_vitb14", memory=1000, points=5000, register=True),
    "C3": dict(model="dator", memory=5000, points=5000, register=True),
    "C4": dict(model="dinov2_vitb14", memory=50000, points=0, register=False),
    # BASELINE configs[4] ("full localise: 50 k-instance memory + 100 k-point objects, 8 GPUs"): a one-GPU SLICE of it -- the clouds,
    # resident features and evaluation grid of this rank's instance range (clouds sharded, routed registration; at one rank every
    # instance is local), 100 000-point objects, 8 frames per step.  --memory sets the slice: 6 250 = one GPU's share of 50 000 instances
    # (625 M points: 105 GB of compact resident features + 10 GB of clouds); the default keeps the run to a few minutes.
    "C5": dict(model="dinov2_vitb14", memory=1024, points=100000, register=True, frames=8, shard_clouds=True, compact=True, arena_gb=48.0,
               steps=3, warmup=1),
}


def smooth_image(rng, h=224, w=224, ch=3):
    """a distinct low-frequency pattern per instance (random-init ViTs barely separate white noise)"""
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    img = np.zeros((h, w, ch), dtype=np.float32)
    for c in range(ch):
        acc = np.zeros((h, w), dtype=np.float32)
        for _ in range(4):
            fx, fy = rng.uniform(0.5, 6, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            acc += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
        img[:, :, c] = acc
    img = (img - img.min()) / (img.max() - img.min() + 1e-9)
    return img


def host_cores():
    """host cores this process may use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes give one GPU a 16-core share)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


class Crops:
    """Per-instance base crops (generated on first use) and their noisy variants (memory views and query crops), for the ViT encoders
    (224^2 RGB u8) and for DATOR (256x128 RGB u8 + 64x32 float depth in metres)."""

    def __init__(self, model, seed):
        self.dator = model == "dator"
        self.hw = (256, 128) if self.dator else (224, 224)
        self.seed = seed
        self._base, self._depth = {}, {}

    def base(self, k):
        if k not in self._base:
            r = np.random.default_rng([self.seed, int(k)])
            self._base[k] = smooth_image(r, *self.hw)
            if self.dator:
                self._depth[k] = (0.3 + 7.7 * smooth_image(r, 64, 32, 1)[:, :, 0]).astype(np.float32)
        return self._base[k]

    def variants(self, ids, r, device):
        import torch
        out = np.empty((len(ids),) + self.hw + (3,), dtype=np.uint8)
        for i, k in enumerate(ids):
            b = self.base(k)
            v = b + r.normal(0, 0.03, size=b.shape).astype(np.float32)
            out[i] = np.clip(v * 255.0, 0, 255).astype(np.uint8)
        rgb = torch.from_numpy(out).to(device)
        if not self.dator:
            return rgb
        d = np.stack([self._depth[k] for k in ids]) + r.normal(0, 0.02, size=(len(ids), 64, 32)).astype(np.float32)
        return (rgb, torch.from_numpy(d.astype(np.float32)).to(device))


class VarCrops:
    """Crops of varying size (H, W ~ U{64..400}): the instance's low-frequency pattern evaluated on the crop's own
    pixel grid + noise, so that every size goes through the resize / centre-crop kernel (`ibl_preprocess_crops`) like a detector's
    bounding-box crop does."""

    def __init__(self, seed, lo=64, hi=400):
        self.seed, self.lo, self.hi = seed, lo, hi
        self._par = {}

    def params(self, k):
        if k not in self._par:
            r = np.random.default_rng([self.seed, int(k)])
            self._par[k] = (r.uniform(0.5, 6, size=(3, 4, 2)), r.uniform(0, 2 * np.pi, size=(3, 4)), r.uniform(0.3, 1.0, size=(3, 4)))
        return self._par[k]

    def make(self, k, r, size=None):
        h, w = size if size is not None else (int(r.integers(self.lo, self.hi + 1)), int(r.integers(self.lo, self.hi + 1)))
        f, ph, amp = self.params(k)
        yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
        img = np.empty((h, w, 3), dtype=np.float32)
        for c in range(3):
            acc = sum(amp[c, i] * np.sin(2 * np.pi * (f[c, i, 0] * xx + f[c, i, 1] * yy) + ph[c, i]) for i in range(4))
            img[:, :, c] = 0.5 + acc / (2 * amp[c].sum())
        img += r.normal(0, 0.03, size=img.shape).astype(np.float32)
        return np.clip(img * 255.0, 0, 255).astype(np.uint8)

    def variants_host(self, ids, r):
        return [self.make(k, r) for k in ids]

    def variants(self, ids, r, device):
        from ibloc_amd.vit import PackedCrops
        return PackedCrops(self.variants_host(ids, r), device)


def make_encoder(model, device):
    from ibloc_amd import vit as V
    if model == "dator":
        from ibloc_amd import dator as D
        return D.DatorEncoder(D.random_stream_weights(20), D.random_stream_weights(21), D.random_head_weights(22), device=device), 128
    cfg = V.CONFIGS[model]
    return V.VitEncoder(cfg, V.random_weights(cfg, 20), device=device), cfg.out_dim


def embed_memory(args, world, world_size, enc, crops, rng, device):
    """memory embeddings = the encoder's embeddings of E noisy views of every instance"""
    embed_ids = np.arange(args.memory)
    if not args.register and args.memory > 20000:
        seen = set()
        for r in range(world_size):
            rr = np.random.default_rng(1000 + r)
            for step in range(args.warmup + args.steps):
                for _ in range(args.frames):
                    seen.update(world.make_frame(rr, q=args.q, pts_per_object=args.points, with_clouds=False)["ids"])
        embed_ids = np.array(sorted(seen))
    mem_emb = world.embeddings.astype(np.float32).copy()
    mem_emb /= np.linalg.norm(mem_emb, axis=-1, keepdims=True)
    ids_all = np.repeat(embed_ids, args.views)
    got = []
    for i in range(0, len(ids_all), 256):
        got.append(enc.embed(crops.variants(ids_all[i:i + 256], rng, device)).cpu().numpy())
    got = np.concatenate(got).reshape(len(embed_ids), args.views, -1)
    if len(embed_ids) < args.memory:
        mem_emb = mem_emb * np.linalg.norm(got, axis=-1).mean()           # the scale of the encoder's (un-normalised) outputs
    mem_emb[embed_ids] = got
    return mem_emb


def build_workload(args, rank, world_size, device, reuse=None, n_steps=None):
    """reuse = (encoder, dim, memory embeddings) of a workload built before: the same memory with another object spacing (the embeddings
    of the synthetic instances do not depend on where the objects stand); n_steps: query batches to generate (default warmup + steps)"""
    import torch
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    from ibloc_amd.synth import SynthWorld

    t0 = time.time()
    n_steps = args.warmup + args.steps if n_steps is None else n_steps
    if reuse is not None:
        enc, dim, mem_emb_reused, comm_reused = reuse
    else:
        enc, dim = make_encoder(args.model, device)
    world = SynthWorld(args.memory, pts_per_object=max(args.points, 16), E=args.views, D=dim, seed=21,
                       sample_points=args.register, spacing=args.spacing)
    rng = np.random.default_rng(21)
    crops = VarCrops(21) if args.var_crops else Crops(args.model, 21)
    # query batches first (distinct per step and per rank): they tell which instances are ever looked at
    frames = []
    frng = np.random.default_rng(1000 + rank)
    for step in range(n_steps):
        frames.append([world.make_frame(frng, q=args.q, pts_per_object=args.points, with_clouds=args.register) for _ in range(args.frames)])
    # memory embeddings = the encoder's embeddings of E noisy views of every instance.  An embedding-only memory of 50 000
    # instances (config C4) would need 200 000 synthetic crops that no query ever looks at: there only the instances that occur in
    # a query frame (of any rank) are embedded, the others keep the generator's random unit embeddings -- throughput is unaffected
    # (the match streams every row either way), the accuracy fields then describe the embedded instances only
    if reuse is not None:
        mem_emb = mem_emb_reused
    else:
        mem_emb = embed_memory(args, world, world_size, enc, crops, rng, device)
    ctx = RegContext(int(args.arena_gb * (1 << 30)))
    comm = None
    if reuse is not None:
        comm = comm_reused
    elif args.shard_memory and args.comm == "rccl":          # the library's own RCCL communicator instead of torch.distributed's group
        from ibloc_amd.parallel import RcclComm
        with stdout_to_stderr():
            comm = RcclComm() if world_size > 1 else RcclComm.single()
    shard = (rank, world_size) if (args.shard_memory and (world_size > 1 or comm is not None)) or args.shard_clouds else None
    # --shard-clouds: this rank keeps the clouds / cached features / evaluation grid of its instance range only; registration jobs are
    # routed between the ranks (ibloc_amd/routing.py).  At one rank it runs the routed path with every instance local.
    mem = MemoryShard(ctx, list(mem_emb), world.points if args.register else None, colors=world.colors if args.register else None,
                      device=device, shard=shard, shard_clouds=args.shard_clouds, compact_features=args.compact_features)
    # host cores are shared by the ranks of the node: the assignment search takes its share, at most 16 threads
    cores = host_cores()
    eng = LocaliseEngine(mem, enc, assign_threads=max(1, min(16, cores // max(1, world_size))), rows_cap=args.frames * 7, comm=comm)
    # query batches, device resident
    batches = []
    for fl in frames:
        clouds, ints, crop_ids, qs, poses, ids = [], [], [], [], [], []
        for f in fl:
            for (p, c) in f["clouds"]:
                clouds.append(p)
                ints.append(intensity_from_colors(c))
            crop_ids += f["ids"]
            qs.append(len(f["ids"]))
            poses.append(f["pose"])
            ids.append(f["ids"])
        det = CloudBatch.from_numpy(clouds, ints, device=device) if args.register else None
        if args.var_crops:
            from ibloc_amd.vit import PackedCrops
            host = crops.variants_host(crop_ids, frng)
            batches.append(dict(det=det, crops=PackedCrops(host, device), crops_host=host, qs=qs, poses=poses, ids=ids))
        else:
            batches.append(dict(det=det, crops=crops.variants(crop_ids, frng, device), qs=qs, poses=poses, ids=ids))
    if rank == 0:
        print(f"[bench] setup {time.time() - t0:.1f}s: M={args.memory} E={args.views} pts={args.points} "
              f"frames/step={args.frames} Q={args.q} model={args.model}", file=sys.stderr)
    return eng, batches, world, mem_emb


def cpu_baseline(args, world, mem_emb, batch, n_frames=1):
    """The oracle (CPU restatement, `kind: port`) on a bounded sample: n_frames frames of the same workload.
    Assignment uses the host search of the library (the reference's own O(M^3) volume cannot run at M >= 1000)."""
    from ibloc_amd import vit as V
    from ibloc_amd.assign import assign_batch
    from ibloc_amd import preprocess as pp
    from oracle import match_oracle as mo
    from oracle import reg_oracle as ro
    from oracle import vit_oracle as vo

    import torch
    threads = min(16, host_cores())                 # the GPU box gives one GPU's share of 16 host cores
    torch.set_num_threads(threads)
    from oracle.clib import lib as olib
    olib.oracle_set_threads(threads)
    if args.model == "dator":
        from ibloc_amd import dator as D
        from oracle import dator_oracle as do
        wr, wd, wh = D.fold_lora(D.random_stream_weights(20)), D.fold_lora(D.random_stream_weights(21)), D.random_head_weights(22)
        rgb = batch["crops"][0].cpu().numpy()
        dep = batch["crops"][1].cpu().numpy()

        def embed(lo, hi):
            return do.embed(wr, wd, wh, list(rgb[lo:hi]), list(dep[lo:hi]))
    else:
        cfg = V.CONFIGS[args.model]
        w = V.random_weights(cfg, 20)
        crops = batch["crops_host"] if "crops_host" in batch else batch["crops"].cpu().numpy()

        def embed(lo, hi):
            return vo.embed_crops(w, cfg, pp.RECIPES[cfg.recipe], list(crops[lo:hi]))
    det = batch["det"]
    if det is not None:
        pts = det.pts4.cpu().numpy()
        off = det.seg_off_host
    mem_n = mo.normalize_rows(mem_emb.reshape(-1, mem_emb.shape[-1]))
    emb_off = (np.arange(args.memory + 1) * args.views).astype(np.int32)
    t0 = time.time()
    row = 0
    for f in range(n_frames):
        q = batch["qs"][f]
        e = embed(row, row + q)
        sims = mo.closest_similarity(mo.normalize_rows(e), mem_n, emb_off)
        aug = np.ones((1, 7, args.memory + 1), dtype=np.float16)          # (the library's assignment search takes 7-row frames)
        aug[0, :q, :-1] = sims
        assns = assign_batch(aug, [q], 4, 1)[0]
        if det is not None:
            cleaned, cint = [], []
            for d in range(q):
                p = pts[off[row + d]:off[row + d + 1]]
                k = ro.radius_outlier(p[:, :3], 0.05, 8)
                cleaned.append(p[k, :3])
                cint.append(np.repeat(p[k, 3:4], 3, axis=1))
            ro.localise_from_assignments(cleaned, cint, world.points, world.colors, assns, 0.05, 1.5, 1.5, seed=args.seed)
        row += q
    dt = time.time() - t0
    return n_frames / dt, dt, threads


class stdout_to_stderr:
    """File-descriptor-level redirect of stdout to stderr for its block: librccl prints a version banner on STDOUT when a communicator is
    created, and the contract of this script is ONE JSON line on rank 0's stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run` (this process has
    made no HIP call and never will: a process that initialised the GPU must not exec or fork GPU work), relay their output -- rank 0
    prints the JSON line -- and exit with the child's code, so a failed rank fails the run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc != 0:
        print(f"[bench] the {n}-rank run failed (exit code {rc})", file=sys.stderr)
    sys.exit(rc)


def rot_err(R_est, R_gt):
    c = (np.trace(R_est.T @ R_gt) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; C5: 3)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 3; C5: 1)")
    ap.add_argument("--config", default="T", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=None, help="query frames per step and per GPU (default 32; C5: 8)")
    ap.add_argument("--memory", type=int, default=None)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--q", type=int, default=None, help="detections per frame (default 7; C1: 4)")
    ap.add_argument("--model", default=None)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--arena-gb", type=float, default=None, help="scratch arena of the registration context (default 24; C5: 48)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--lanes", type=int, default=0, help="(experiment) N concurrent whole-batch lanes instead of the two-stage pipeline")
    ap.add_argument("--sequential", action="store_true", help="run the steps back to back instead of pipelined")
    ap.add_argument("--shard-memory", action="store_true", help="shard the embedding memory by instance range over the ranks "
                    "(per-shard candidate top-k + RCCL all-gather) instead of replicating it")
    ap.add_argument("--shard-clouds", action="store_true", help="also shard the memory clouds by instance range: registration jobs run at "
                    "the owner of their targets or fetch the instances they miss, whole-memory evaluation is reduced over the ranks")
    ap.add_argument("--comm", default=None, choices=["torch", "rccl"], help="transport of the sharded layout's collectives: "
                    "torch.distributed's nccl (= RCCL) group (default at N > 1), or the library's own RCCL communicator (ibl_comm_*; default "
                    "at one rank)")
    ap.add_argument("--layout", default="auto", choices=["auto", "sharded", "replicated"],
                    help="N > 1: `sharded` (default) = embedding memory sharded by instance range + RCCL exchange of the per-shard top-k "
                    "candidate lists (north star); `replicated` = every rank holds the whole memory, no data-path collective")
    ap.add_argument("--compact-features", action="store_true", help="keep the memory's resident instance features without their fp16 search "
                    "operands (168 instead of 264 bytes per point; the feature search converts while it stages)")
    ap.add_argument("--spacing", type=float, default=2.5, help="grid spacing of the synthetic memory's objects in metres (2.5: separated "
                    "objects; ~0.7: adjacent objects whose neighbourhoods overlap, so cross-instance features are recomputed)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of K steps each: `value` is the first (the contract's), value_min / "
                    "value_max the spread over all of them")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false", help="skip the extra region that copies every step's inputs from pinned host memory")
    ap.add_argument("--adjacent-spacing", type=float, default=None, help="spacing of the second, adjacent-objects world measured after the "
                    "timed region and reported as value_adjacent (default 0.7 for config T, 0 = skip for the other configs)")
    ap.add_argument("--ransac-budget", type=int, default=100000, help="hypotheses per job of the fixed-budget RANSAC figure (0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                              # does not return
    preset = CONFIGS[args.config]
    for k in ("model", "memory", "points"):
        if getattr(args, k) is None:
            setattr(args, k, preset[k])
    args.register = preset["register"] and args.points > 0
    if args.q is None:
        args.q = preset.get("q", 7)
    for k, dflt in (("steps", 20), ("warmup", 3), ("frames", 32), ("arena_gb", 24.0)):
        if getattr(args, k) is None:
            setattr(args, k, preset.get(k, dflt))
    if preset.get("shard_clouds"):
        args.shard_clouds = True
    if preset.get("compact"):
        args.compact_features = True
    args.var_crops = bool(preset.get("var_crops", False))
    if args.adjacent_spacing is None:
        args.adjacent_spacing = 0.7 if args.config == "T" else 0.0

    import torch
    import torch.distributed as dist
    from scipy.spatial.transform import Rotation

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.layout == "sharded" or (args.layout == "auto" and world_size > 1 and not args.shard_clouds):
        args.shard_memory = True
    if args.comm is None:
        # N > 1: the collectives of the sharded layout go through torch.distributed's nccl group (= RCCL over xGMI) unless --comm rccl asks
        # for the library's own communicator (ibl_comm_*): the latter has run with one rank only (tests/test_gpu_multirank.py is its
        # two-GPU test), and a first multi-GPU run should not depend on it.  One rank: the library's communicator (measured).
        # (the one-GPU rehearsal with every rank on device 0 cannot use RCCL at all: it refuses two ranks per device)
        args.comm = "rccl" if args.shard_memory and world_size == 1 and os.environ.get("IBL_BENCH_SHARE_GPU", "") != "1" else "torch"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # rehearsal of the multi-rank flow on a one-GPU box (not a measurement): IBL_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # uses gloo for the barrier / max-over-ranks (RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("IBL_BENCH_SHARE_GPU", "") == "1"
    if share_gpu:
        local_rank = 0
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        with stdout_to_stderr():             # (the first collective creates the RCCL communicator: its banner goes to stderr)
            dist.init_process_group("gloo" if share_gpu else "nccl")
            t0_ = torch.zeros(1, device="cpu" if share_gpu else f"cuda:{local_rank}")
            dist.all_reduce(t0_)
            if not share_gpu:
                torch.cuda.synchronize()
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    eng, batches, world, mem_emb = build_workload(args, rank, world_size, device)

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kw = dict(fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, register=args.register)

    def run_step(b, timings=None):
        return eng.localise_batch(b["det"], b["qs"], crops=b["crops"], seed=args.seed, timings=timings, **kw)

    def stream_items(lo, hi):
        return [dict(det=b["det"], q_per_frame=b["qs"], crops=b["crops"], seed=args.seed) for b in batches[lo:hi]]

    def run_many(lo, hi, timings=None):
        if args.sequential:
            return (run_step(b, timings) for b in batches[lo:hi])
        if args.lanes:
            return eng.localise_concurrent(stream_items(lo, hi), workers=args.lanes, worker_arena_bytes=12 << 30, **kw)
        return eng.localise_stream(stream_items(lo, hi), **kw)

    for _ in run_many(0, args.warmup):
        pass
    barrier()
    from ibloc_amd import prof
    prof.reset(enable=os.environ.get("IBL_BENCH_NOPROF", "") == "")
    timings = {}
    t0 = time.perf_counter()
    # A step = one batch of frames through embed -> match -> assign -> register -> evaluate -> pose.  Consecutive steps are
    # pipelined (LocaliseEngine.localise_stream: embed + match of step k + 1 on a second stream while step k is assigned and
    # registered); every step's work starts and ends inside the timed region.  --sequential runs them back to back.
    all_res = list(run_many(args.warmup, args.warmup + args.steps, timings))
    barrier()
    dt = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_frames = args.frames * args.steps * world_size
    value = total_frames / dt
    roof_live = prof.roofline(None)                       # before the untimed stage-timing step below adds launches

    def timed_region(run):
        """one more region of exactly K steps on the same batches, bracketed like the first: frames/s over all ranks"""
        barrier()
        prof.reset(enable=False)
        t1 = time.perf_counter()
        for _ in run():
            pass
        barrier()
        d = time.perf_counter() - t1
        if world_size > 1:
            tt = torch.tensor([d], dtype=torch.float64, device="cpu" if share_gpu else device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return total_frames / d

    # the 20-step region lasts under a second: two repeats of the same region give the run-to-run spread of `value`
    repeats = [value] + [timed_region(lambda: run_many(args.warmup, args.warmup + args.steps)) for _ in range(max(0, args.repeats - 1))]
    # ... and the same K steps with every step's inputs (crops, detected clouds, segment offsets) copied from PINNED HOST memory inside
    # the timed region, on the stream the step runs on: the PCIe-inclusive rate (never `value`: the contract counts HBM-resident inputs)
    value_h2d = None
    if args.h2d and not args.lanes:
        def pin(t):
            return t.cpu().pin_memory()

        host = []
        for b in batches[args.warmup:args.warmup + args.steps]:
            c = b["crops"]
            hb = dict(qs=b["qs"], det_host=None, crops_host=None)
            if b["det"] is not None:
                hb["det_host"] = (pin(b["det"].pts4), b["det"].seg_off_host)
            if isinstance(c, tuple):
                hb["crops_host"] = tuple(pin(x) for x in c)
            elif isinstance(c, torch.Tensor):
                hb["crops_host"] = pin(c)
            else:                                           # PackedCrops: one byte string + shapes
                hb["crops_host"] = (pin(c.src), c.shapes, c.offs)
            host.append(hb)
        h2d_bytes = 0

        def upload(hb):
            nonlocal h2d_bytes
            from ibloc_amd.registration import CloudBatch
            det = None
            if hb["det_host"] is not None:
                p4, off = hb["det_host"]
                det = CloudBatch(p4.to(device, non_blocking=True), off)
                h2d_bytes += p4.numel() * 4
            ch = hb["crops_host"]
            if isinstance(ch, tuple) and len(ch) == 3 and not isinstance(ch[1], torch.Tensor):
                from ibloc_amd.vit import PackedCrops
                crops_d = PackedCrops(None, _src=ch[0].to(device, non_blocking=True), _shapes=ch[1], _offs=ch[2])
                h2d_bytes += ch[0].numel()
            elif isinstance(ch, tuple):
                crops_d = tuple(x.to(device, non_blocking=True) for x in ch)
                h2d_bytes += sum(x.numel() * x.element_size() for x in ch)
            else:
                crops_d = ch.to(device, non_blocking=True)
                h2d_bytes += ch.numel() * ch.element_size()
            return dict(det=det, q_per_frame=hb["qs"], crops=crops_d, seed=args.seed)

        def run_h2d():
            if args.sequential:
                return (eng.localise_batch(u["det"], u["q_per_frame"], crops=u["crops"], seed=args.seed, **kw) for u in map(upload, host))
            return eng.localise_stream((upload(hb) for hb in host), **kw)       # lazy: a batch is uploaded when the pipeline pulls it

        for _ in run_h2d():                                 # one untimed pass: pinned staging and allocator warm
            pass
        h2d_bytes = 0
        value_h2d = {"value": timed_region(run_h2d), "h2d_bytes_per_step": h2d_bytes / max(1, args.steps),
                     "what": "the same K steps with each step's crops + detected clouds copied from pinned host memory inside the timed region"}
    roof_iso = None
    stage_roofs = []
    if not args.sequential:                               # per-stage device times of one step run back to back, outside the timed region
        # (averaged over up to four different batches: a batch with a job that never reaches RANSAC's confidence exit walks all 4 M
        # hypotheses, or whose targets lie far apart, can double its "register" time -- one such batch is not the typical step)
        n_extra = max(1, min(4, args.steps))
        prof.reset(enable=os.environ.get("IBL_BENCH_NOPROF", "") == "")
        for i in range(n_extra):
            run_step(batches[args.warmup + args.steps - 1 - i], timings=timings)
        torch.cuda.synchronize()
        for k in list(timings):
            if isinstance(timings[k], float):
                timings[k] /= n_extra
        if os.environ.get("IBL_TIMING") or os.environ.get("IBL_BENCH_DEBUG"):
            print("[bench] stage timings of the extra steps:", {k: v for k, v in timings.items() if isinstance(v, float)}, file=sys.stderr)
        roof_iso = prof.roofline(None)                    # the same GEMM launches with nothing else on the device
        stage_roofs = prof.stage_rooflines() if args.register else []     # one entry per registration stage, same steps
        for e in stage_roofs:
            e["ms_per_step"] = e.pop("ms") / n_extra
            e["work_per_step"] = e.pop("work") / n_extra
        # stage B is one library call (ibl_register_evaluate_batch): its split comes from the library's stage timer
        for key, name in (("outlier", "outlier (a9)"), ("det_features", "normals+FPFH (a11)"), ("feature_match", "feature match (a12)"),
                          ("ransac", "RANSAC (a12)"), ("icp", "coloured ICP (a12)"), ("evaluate", "evaluate (a13)")):
            for e in stage_roofs:
                if e["stage"] == name:
                    timings[key] = e["ms_per_step"]
        # SURVEY 8d's fixed-budget RANSAC figure: H hypotheses per job with the confidence exit off (never the product setting)
        if args.register and args.ransac_budget > 0 and not args.shard_clouds:
            prof.reset(enable=True)
            b = batches[args.warmup + args.steps - 1]
            eng.localise_batch(b["det"], b["qs"], crops=b["crops"], seed=args.seed, ransac_max_iter=args.ransac_budget,
                               ransac_fixed_budget=True, **kw)
            torch.cuda.synchronize()
            ms, hyp, n = prof.read(prof.ST_RANSAC)
            if n and ms > 0:
                stage_roofs.append({"stage": "RANSAC fixed budget (a12)", "bound": "valu", "achieved": hyp / (ms * 1e-3) / 1e6, "peak": None,
                                    "unit": "Mhyp/s", "frac": None, "ms_per_step": ms, "calls": n, "work_per_step": hyp,
                                    "work_unit": f"H = {args.ransac_budget} hypotheses per job, confidence exit off, {int(hyp // args.ransac_budget)} jobs; "
                                    "each = Philox draw + edge-length test, survivors (~1 %) get the 3-point Kabsch + distance check + scoring on "
                                    "all correspondences (the checkers prune before scoring, as in Open3D)"})
            prof.reset(enable=False)

    def accuracy(results, bs):
        """accuracy signals against the generator's ground truth (outside the timed regions)"""
        n_frames = n_assn_ok = n_ok = n_ok_given = n_any = 0
        clean_pts = []
        for res, b in zip(results, bs):
            for f, r in enumerate(res):
                n_frames += 1
                ids = b["ids"][f]
                if r.best < 0:
                    continue
                chosen = r.assignments[r.best]
                good = all(ids[d] == m for d, m in chosen)
                n_any += int(any(all(ids[d] == m for d, m in a) for a in r.assignments))
                n_assn_ok += int(good)
                if args.register:
                    P = b["poses"][f]
                    terr = np.linalg.norm(r.pose_corrected[:3] - P[:3, 3])
                    rerr = rot_err(Rotation.from_quat(r.pose_corrected[3:]).as_matrix(), P[:3, :3])
                    ok = terr < 0.6 and rerr < 0.3            # the reference's success rule, tum_localisation_trial.py:274
                    n_ok += int(ok)
                    n_ok_given += int(ok and good)
                    clean_pts.append(r.n_clean / max(1, len(ids)))
        return n_frames, n_assn_ok, n_ok, n_ok_given, n_any, clean_pts

    n_frames, n_assn_ok, n_ok, n_ok_given, n_any, clean_pts = accuracy(all_res, batches[args.warmup:args.warmup + args.steps])

    # ---- the same workload with ADJACENT objects: the default world keeps its objects 2.5 m apart, so no instance lies
    # inside another one's feature neighbourhood and every job side is served by the resident instance features -- the best case.  On a
    # 0.7 m grid (a desk) the features of every multi-instance job side are recomputed on the concatenation, as the reference does for
    # every job.  Same memory embeddings, same encoder, K steps timed the same way, reported beside `value`.
    adjacent = None
    cpu_batch = batches[args.warmup]                       # (the CPU baseline leg below samples the default world)
    if args.adjacent_spacing > 0 and args.register and abs(args.spacing - args.adjacent_spacing) > 1e-9:
        import copy
        t_adj = time.time()
        a2 = copy.copy(args)
        a2.spacing = args.adjacent_spacing
        comm0 = eng.exchange.comm if eng.exchange is not None else None
        enc0 = eng.encoder
        eng.close()
        eng.memory.close()
        eng.ctx.close()                                    # the first world's arena (24 GB) and resident features go before the second is built
        eng.memory = None
        del batches
        torch.cuda.empty_cache()
        eng2, batches2, _, _ = build_workload(a2, rank, world_size, device, reuse=(enc0, mem_emb.shape[-1], mem_emb, comm0), n_steps=2 + args.steps)
        eng = eng2                                         # (the communicator is released through this engine below)
        items2 = [dict(det=b["det"], q_per_frame=b["qs"], crops=b["crops"], seed=args.seed) for b in batches2]

        def run2(lo, hi):
            if args.sequential:
                return (eng2.localise_batch(b["det"], b["q_per_frame"], crops=b["crops"], seed=args.seed, **kw) for b in items2[lo:hi])
            return eng2.localise_stream(items2[lo:hi], **kw)
        for _ in run2(0, 2):
            pass
        barrier()
        prof.reset(enable=False)
        t1 = time.perf_counter()
        res2 = list(run2(2, 2 + args.steps))
        barrier()
        d2 = time.perf_counter() - t1
        if world_size > 1:
            tt = torch.tensor([d2], dtype=torch.float64, device="cpu" if share_gpu else device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d2 = float(tt.item())
        tm2 = {}
        eng2.localise_batch(batches2[-1]["det"], batches2[-1]["qs"], crops=batches2[-1]["crops"], seed=args.seed, timings=tm2, **kw)
        torch.cuda.synchronize()
        nf, na, nok, nokg, nany, cp = accuracy(res2, batches2[2:2 + args.steps])
        adjacent = {"value": total_frames / d2, "unit": "query-frames/s", "ms_per_step": d2 / args.steps * 1e3, "object_spacing_m": args.adjacent_spacing,
                    "steps": args.steps, "assignment_correct_rate": na / max(1, nf), "localised_0.6m_0.3rad": nok / max(1, nf),
                    "registered_given_correct_assignment": nokg / max(1, na), "det_points_after_outlier_mean": float(np.mean(cp)) if cp else None,
                    "feature_reuse_last_step": tm2.get("reuse"),
                    "what": "the same memory, encoder and step on objects %.2f m apart: neighbouring instances lie inside each other's feature "
                            "neighbourhoods, so the features of every multi-instance job side are recomputed on the concatenation "
                            "(feature_reuse: points served by resident features, points recomputed, recomputed groups, job sides, distinct "
                            "matching pairs, pair uses)" % args.adjacent_spacing}
        if rank == 0:
            print(f"[bench] adjacent-objects region {time.time() - t_adj:.1f}s", file=sys.stderr)

    if rank == 0:
        roof = roof_live
        if roof is not None and roof_iso is not None:
            roof["note"] = ("pipelined steps: these launches share the device with the registration kernels of the previous step, so the "
                            "duration of a launch is not the kernel's own speed; roofline_isolated times the same launches alone; "
                            "traffic: PMC counters cannot be read in-process -- it is the FETCH_SIZE / WRITE_SIZE result of the committed "
                            "rocprofv3 --pmc passes of this build (tools/profile_pmc.sh; matched by the kernel source's hash)")
        # roofline.traffic: the FETCH_SIZE / WRITE_SIZE figure of THIS build -- tools/profile_pmc.sh stamps profiles/r04/gemm_pmc.json with
        # the sha256 of csrc/vit.hip it was measured on; a file measured on other kernel source is not echoed (traffic stays null)
        pmc_file = os.path.join(ROOT, "profiles", "r04", "gemm_pmc.json")
        if roof is not None and os.path.exists(pmc_file):
            try:
                import hashlib
                pm = json.load(open(pmc_file))
                sha = hashlib.sha256(open(os.path.join(ROOT, "instance-based-loc_amd", "csrc", "vit.hip"), "rb").read()).hexdigest()
                if pm.get("source_sha256_vit_hip") == sha:
                    for r_ in (roof, roof_iso):
                        if r_ is not None:
                            r_["traffic"] = pm.get("traffic_bytes_per_launch")
                            r_["traffic_source"] = {"file": "profiles/r04/gemm_pmc.json", "algorithmic_bytes_per_launch": pm.get("algorithmic_bytes_per_launch"),
                                                    "how": pm.get("how"), "source_sha256_vit_hip": sha[:16]}
                else:
                    roof["traffic_source"] = "profiles/r04/gemm_pmc.json was measured on other kernel source than this build's: not reported"
            except (OSError, ValueError):
                pass
        cpu = None
        cpu_frames = args.cpu_frames if args.cpu_frames is not None else (6 if args.register else 24)
        if cpu_frames > 0 and world_size == 1:            # the CPU leg is timed on rank 0 of the single-GPU run only
            v, cdt, threads = cpu_baseline(args, world, mem_emb, cpu_batch, cpu_frames)
            enc_name = "torch-cpu fp32 " + ("DATOR (two TransReID streams + fusion head)" if args.model == "dator" else "ViT")
            reg_name = "C oracle (match, FPFH, RANSAC, coloured ICP, evaluate; OpenMP) + host assignment search" if args.register \
                else "C oracle match + host assignment search"
            cpu = {"value": v, "unit": "query-frames/s", "cores": threads, "kind": "port",
                   "sample": f"{cpu_frames} frame(s) of the same workload, {cdt:.1f} s: {enc_name} + {reg_name}"}
        stages = "FPFH+RANSAC+coloured ICP on %d-pt clouds, whole-memory evaluate" % args.points if args.register else "embed+match+assign only"
        crop_desc = "256x128 RGB + depth crops" if args.model == "dator" else ("crops of 64..400 px (resized on the device)" if args.var_crops else "crops 224^2")
        out = {
            "metric": "query-frames localized/sec (embed+match+register)",
            "value": value,
            "value_min": min(repeats), "value_max": max(repeats), "value_repeats": repeats,
            "value_with_h2d": value_h2d,
            "value_adjacent": adjacent,
            "unit": "query-frames/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 (ViT MFMA operands, f32 accumulate) / f32+f64 (match, registration)",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {args.model} {crop_desc} (Q={args.q}), {args.memory}-instance memory (E={args.views}), {stages}"
                       + ("; synthetic objects of extent U[0.15, 0.6] m on a %.2f m grid (SURVEY 8d's U[0.2, 1.5] m re-scoped: its 5 000 points "
                          "are too sparse to survive the radius-outlier removal)" % args.spacing if args.register else ""),
                       "frames_per_step_per_gpu": args.frames, "memory_instances": args.memory,
                       "points_per_object": args.points, "object_spacing_m": args.spacing,
                       "resident_feature_bytes_per_point": 168 if args.compact_features else 264,
                       "parallelism": f"frames-dp{world_size}" + ("+memory-shard%d(%s)" % (world_size, args.comm) if args.shard_memory else "+memory-replicated")
                       + ("+cloud-shard%d" % world_size if args.shard_clouds else ""),
                       "rccl_ranks": (eng.exchange.comm.world if eng.exchange is not None and eng.exchange.comm is not None
                                      else (world_size if world_size > 1 and not share_gpu else None))},
            # `roofline`: HIP events around every GEMM launch of the timed region.  With pipelined steps the embed stream shares the
            # device with the registration kernels of the previous step, so a launch's duration there is not the kernel's own
            # speed; `roofline_isolated` is the same measurement over the launches of one extra step run alone afterwards.
            "roofline": roof,
            "roofline_isolated": roof_iso,
            # SURVEY 8d "report each stage separately": the registration stages of the same isolated steps (HIP events on the launch
            # stream around each stage's launches x algorithmic work), plus the fixed-budget RANSAC figure
            "roofline_stages": stage_roofs,
            "cpu_baseline": cpu,
            "pipelined_steps": not args.sequential,
            "stage_ms_per_step": {k: v / (args.steps if args.sequential else 1) for k, v in timings.items() if isinstance(v, float)},
            # accuracy signals of rank 0's frames against the generator's ground truth (random-init encoder weights: the embeddings
            # separate the synthetic instances only weakly, so these describe the workload, not the method)
            "det_points_after_outlier_mean": float(np.mean(clean_pts)) if clean_pts else None,
            "assignment_correct_rate": n_assn_ok / max(1, n_frames),
            "assignment_candidates_contain_correct": n_any / max(1, n_frames),
            "localised_0.6m_0.3rad": (n_ok / max(1, n_frames)) if args.register else None,
            "registered_given_correct_assignment": (n_ok_given / max(1, n_assn_ok)) if args.register else None,
            # last step: points whose normals/FPFH/gradients came from the resident instance features, points recomputed in the
            # context of their job (instances within the influence radius of each other), recomputed groups, job sides
            "feature_reuse_last_step": timings.get("reuse"),
        }
        if args.config == "C1":
            # the only numbers that exist for the literal reference code at this scale (measured in the survey container,
            # 8 host cores, no GPU; not reference-published and not part of cpu_baseline, which times this repo's CPU restatement)
            out["reference_cpu_timings_survey6"] = {
                "SimVolume construct + top-k, Q=7, M=20 (utils/similarity_volume.py, numba stubbed)": "1.41 s + 0.30 s per frame",
                "DINOv2-B/14 architecture 224^2, batch 1, fp32 torch-cpu, 8 threads, random weights": "139 ms per crop",
                "source": "SURVEY.md section 6"}
        print(json.dumps(out))
    if eng.exchange is not None and eng.exchange.comm is not None:       # the library's RCCL communicator: released by every rank, before
        torch.cuda.synchronize()                                            # torch's process group goes
        eng.exchange.comm.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
