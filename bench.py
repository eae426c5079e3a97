#!/usr/bin/env python3
"""bench.py -- query-frames localised / second (embed + match + register) on MI355X.

Workload (BASELINE.json configs[1], "C2"): DINOv2 ViT-B/14 crops (224^2, Q = 7 per frame), 1 000-instance
memory (E = 4 stored embeddings each, 5 000-point coloured clouds), FPFH + RANSAC + coloured ICP on the
5k-point clouds of every candidate assignment, whole-memory evaluation.  A "step" is one pass of the hot path
(ObjectMemory.localise body, object_memory.py:911-1131) over one batch of --frames synthetic query frames whose
crops and detected clouds are already resident in HBM.  Consecutive steps are pipelined the way a localisation service runs
(LocaliseEngine.localise_stream: embed + match of step k + 1 on a second HIP stream while step k is assigned and registered;
--sequential runs them back to back); all work of the K timed steps happens inside the timed region.  Synthetic data and
seeded random-init weights (no datasets / checkpoints offline).

    python bench.py --gpus 1 --steps 40 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: query frames are independent (tum_localisation_trial.py:215 has no cross-frame state), so each
rank localises its own frames against a replicated memory -- no data-path collective, "scaling": "weak".
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


def smooth_image(rng, size=224):
    """a distinct low-frequency RGB pattern per instance (random-init ViTs barely separate white noise)"""
    yy, xx = np.meshgrid(np.linspace(0, 1, size), np.linspace(0, 1, size), indexing="ij")
    img = np.zeros((size, size, 3), dtype=np.float32)
    for c in range(3):
        acc = np.zeros((size, size), dtype=np.float32)
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


def build_workload(args, rank, device):
    import torch
    from ibloc_amd import vit as V
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    from ibloc_amd.synth import SynthWorld

    t0 = time.time()
    cfg = V.CONFIGS[args.model]
    enc = V.VitEncoder(cfg, V.random_weights(cfg, 20), device=device)
    world = SynthWorld(args.memory, pts_per_object=args.points, E=args.views, D=cfg.out_dim, seed=21)
    rng = np.random.default_rng(21)
    # per-instance base crop; memory views and query crops are noisy variants of it
    base = np.stack([smooth_image(rng) for _ in range(args.memory)])              # (M, 224, 224, 3) in [0, 1]

    def variants(ids, r):
        out = np.empty((len(ids), 224, 224, 3), dtype=np.uint8)
        for i, k in enumerate(ids):
            v = base[k] + r.normal(0, 0.03, size=base[k].shape).astype(np.float32)
            out[i] = np.clip(v * 255.0, 0, 255).astype(np.uint8)
        return out

    mem_emb = []
    ids_all = np.repeat(np.arange(args.memory), args.views)
    for i in range(0, len(ids_all), 256):
        crops = torch.from_numpy(variants(ids_all[i:i + 256], rng)).to(device)
        mem_emb.append(enc.embed(crops).cpu().numpy())
    mem_emb = np.concatenate(mem_emb).reshape(args.memory, args.views, -1)
    ctx = RegContext(int(args.arena_gb * (1 << 30)))
    mem = MemoryShard(ctx, list(mem_emb), world.points, colors=world.colors, device=device)
    # host cores are shared by the ranks of the node: the assignment search takes its share, at most 16 threads
    cores = host_cores()
    eng = LocaliseEngine(mem, enc, assign_threads=max(1, min(16, cores // max(1, int(os.environ.get("WORLD_SIZE", "1"))))))
    # query batches (distinct per step and per rank), device resident
    batches = []
    frng = np.random.default_rng(1000 + rank)
    for step in range(args.warmup + args.steps):
        clouds, ints, crop_ids, qs, poses, ids = [], [], [], [], [], []
        for _ in range(args.frames):
            f = world.make_frame(frng, q=args.q, pts_per_object=args.points)
            for (p, c) in f["clouds"]:
                clouds.append(p)
                ints.append(intensity_from_colors(c))
            crop_ids += f["ids"]
            qs.append(len(f["ids"]))
            poses.append(f["pose"])
            ids.append(f["ids"])
        det = CloudBatch.from_numpy(clouds, ints, device=device)
        crops = torch.from_numpy(variants(crop_ids, frng)).to(device)
        batches.append(dict(det=det, crops=crops, qs=qs, poses=poses, ids=ids))
    if rank == 0:
        print(f"[bench] setup {time.time() - t0:.1f}s: M={args.memory} E={args.views} pts={args.points} "
              f"frames/step={args.frames} Q={args.q} model={args.model}", file=sys.stderr)
    return eng, batches, world, mem_emb


def run_step(eng, b, args, timings=None):
    return eng.localise_batch(b["det"], b["qs"], crops=b["crops"], fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5,
                              fpfh_local_dist_factor=1.5, seed=args.seed, timings=timings)


def cpu_baseline(args, world, mem_emb, batch, n_frames=1):
    """The oracle (CPU restatement, `kind: port`) on a bounded sample: n_frames frames of the same workload.
    Assignment uses the host search of the library (the reference's own O(M^3) volume cannot run at M = 1000)."""
    import dataclasses

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
    cfg = V.CONFIGS[args.model]
    w = V.random_weights(cfg, 20)
    crops = batch["crops"].cpu().numpy()
    det = batch["det"]
    pts = det.pts4.cpu().numpy()
    off = det.seg_off_host
    mem_n = mo.normalize_rows(mem_emb.reshape(-1, mem_emb.shape[-1]))
    emb_off = (np.arange(args.memory + 1) * args.views).astype(np.int32)
    t0 = time.time()
    row = 0
    for f in range(n_frames):
        q = batch["qs"][f]
        e = vo.embed_crops(w, cfg, pp.RECIPES[cfg.recipe], list(crops[row:row + q]))
        sims = mo.closest_similarity(mo.normalize_rows(e), mem_n, emb_off)
        aug = np.ones((1, 7, args.memory + 1), dtype=np.float16)
        aug[0, :q, :-1] = sims
        assns = assign_batch(aug, [q], 4, 1)[0]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="query frames per step and per GPU")
    ap.add_argument("--memory", type=int, default=1000)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--q", type=int, default=7)
    ap.add_argument("--model", default="dinov2_vitb14")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--arena-gb", type=float, default=24.0)
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--lanes", type=int, default=0, help="(experiment) N concurrent whole-batch lanes instead of the two-stage pipeline")
    ap.add_argument("--sequential", action="store_true", help="run the steps back to back instead of pipelined")
    ap.add_argument("--profile-kernel", default="", help="(internal) name of the kernel the roofline is reported for")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    eng, batches, world, mem_emb = build_workload(args, rank, device)

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kw = dict(fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)

    def stream_items(lo, hi):
        return [dict(det=b["det"], q_per_frame=b["qs"], crops=b["crops"], seed=args.seed) for b in batches[lo:hi]]

    if args.sequential:
        for i in range(args.warmup):
            run_step(eng, batches[i], args)
    else:
        for _ in (eng.localise_concurrent(stream_items(0, args.warmup), workers=args.lanes, worker_arena_bytes=12 << 30, **kw) if args.lanes
                  else eng.localise_stream(stream_items(0, args.warmup), **kw)):
            pass
    barrier()
    from ibloc_amd import prof
    prof.reset(enable=os.environ.get("IBL_BENCH_NOPROF", "") == "")
    timings = {}
    t0 = time.perf_counter()
    ok = 0
    # A step = one batch of frames through embed -> match -> assign -> register -> evaluate -> pose.  Consecutive steps are
    # pipelined (LocaliseEngine.localise_stream: embed + match of step k + 1 on a second stream while step k is assigned and
    # registered); every step's work starts and ends inside the timed region.  --sequential runs them back to back.
    if args.sequential:
        step_results = (run_step(eng, batches[args.warmup + i], args, timings=timings) for i in range(args.steps))
    else:
        items = stream_items(args.warmup, args.warmup + args.steps)
        step_results = eng.localise_concurrent(items, workers=args.lanes, worker_arena_bytes=12 << 30, **kw) if args.lanes \
            else eng.localise_stream(items, **kw)
    for i, res in enumerate(step_results):
        b = batches[args.warmup + i]
        for f, r in enumerate(res):
            P = b["poses"][f]
            ok += int(np.linalg.norm(r.pose_corrected[:3] - P[:3, 3]) < 0.6)
    barrier()
    dt = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_frames = args.frames * args.steps * world_size
    value = total_frames / dt
    roof_live = prof.roofline(None)                       # before the untimed stage-timing step below adds launches
    roof_iso = None
    if not args.sequential:                               # per-stage device times of one step run back to back, outside the timed region
        prof.reset(enable=os.environ.get("IBL_BENCH_NOPROF", "") == "")
        run_step(eng, batches[args.warmup + args.steps - 1], args, timings=timings)
        torch.cuda.synchronize()
        roof_iso = prof.roofline(None)                    # the same GEMM launches with nothing else on the device

    if rank == 0:
        # HBM traffic per GEMM launch: PMC counters cannot be read in-process, so the figure comes from the committed
        # rocprofv3 --pmc passes of this same default command (profiles/r01/gemm_pmc.json); null for other workloads
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01", "gemm_pmc.json")
        if os.path.exists(pmc) and (args.model, args.frames, args.q, args.memory) == ("dinov2_vitb14", 32, 7, 1000):
            traffic = json.load(open(pmc)).get("gemm_traffic_bytes_per_launch")
        roof = roof_live
        for r in (roof, roof_iso):
            if r is not None:
                r["traffic"] = traffic
        if roof is not None and roof_iso is not None:
            roof["note"] = ("pipelined steps: these launches share the device with the registration kernels of the previous step, so the "
                            "duration of a launch is not the kernel's own speed; roofline_isolated times the same launches alone")
        cpu = None
        if args.cpu_frames > 0 and world_size == 1:       # the CPU leg is timed on rank 0 of the single-GPU run only
            v, cdt, threads = cpu_baseline(args, world, mem_emb, batches[args.warmup], args.cpu_frames)
            cpu = {"value": v, "unit": "query-frames/s", "cores": threads,
                   "kind": "port", "sample": f"{args.cpu_frames} frame(s) of the same workload, {cdt:.1f} s: torch-cpu fp32 ViT + "
                   "C oracle (match, FPFH, RANSAC, coloured ICP, evaluate; OpenMP) + host assignment search"}
        out = {
            "metric": "query-frames localized/sec (embed+match+register)",
            "value": value,
            "unit": "query-frames/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 (ViT MFMA) / f32+f64 (match, registration)",
            "data": "synthetic",
            "config": {"workload": f"{'C2' if args.memory == 1000 else 'T' if args.memory == 10000 else 'custom'}: {args.model} crops 224^2 "
                       f"(Q={args.q}), {args.memory}-instance memory (E={args.views}), FPFH+RANSAC+coloured ICP on {args.points}-pt clouds, "
                       "whole-memory evaluate", "frames_per_step_per_gpu": args.frames, "memory_instances": args.memory,
                       "points_per_object": args.points, "parallelism": f"frames-dp{world_size}"},
            # `roofline`: HIP events around every GEMM launch of the timed region.  With pipelined steps the embed stream shares the
            # device with the registration kernels of the previous step, so a launch's duration there is not the kernel's own
            # speed; `roofline_isolated` is the same measurement over the launches of one extra step run alone afterwards.
            "roofline": roof,
            "roofline_isolated": roof_iso,
            "cpu_baseline": cpu,
            "pipelined_steps": not args.sequential,
            "stage_ms_per_step": {k: v / (args.steps if args.sequential else 1) for k, v in timings.items() if isinstance(v, float)},
            "localised_within_0.6m": ok / max(1, args.frames * args.steps),
            # last step: points whose normals/FPFH/gradients came from the resident instance features, points recomputed in the
            # context of their job (instances within the influence radius of each other), recomputed groups, job sides
            "feature_reuse_last_step": timings.get("reuse"),
        }
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
