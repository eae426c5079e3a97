"""Batched localisation engine: embed -> match -> assign -> register -> evaluate -> pose for many query
frames at once.  This is the MI355X restatement of the body of ObjectMemory.localise()
(/root/reference/object_memory/object_memory.py:911-1131); the per-frame facade with the reference's
signature lives in object_memory/object_memory.py.

Two stages per batch of frames:
  A  embed the crops, match them against the memory embeddings on the device, keep per query row the two-ended candidate list
     (csrc/topk.hip), bring the candidates to the host (1.3 KB per row instead of the 2 (M + 1)-byte row) and run the exact
     similarity-volume search on them (csrc/assign.cpp; proved equal to the search on full rows, else redone on the full rows).
     With the embedding memory sharded over ranks (MemoryShard(shard=(rank, world))) the query rows and the per-shard candidate
     lists are all-gathered over RCCL here (parallel.ShardExchange) -- every collective of a step is issued by this stage.
  B  clean the detected clouds, compute their features, register every candidate assignment against the (resident) memory clouds,
     evaluate against the whole memory, select and assemble the pose.
`localise_stream` runs stage A of batch k + 1 on a second HIP stream / host thread while stage B of batch k executes.

Everything numerical runs in libibloc_hip.so; torch is used for device memory, stream handling, `torch.distributed` and
order-preserving boolean compaction only.
"""
import os
import warnings
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import match
from .assign import K_HI, K_LO, assign_batch, assign_candidates
from .parallel import ShardExchange, shard_range, sharded_candidates
from .registration import (CloudBatch, InstanceFeatures, MemGrid, RegContext, evaluate_batch, evaluate_points,
                           instance_features_batch, radius_outlier_batch, register_batch, register_evaluate_batch)

IBL_ST_GRID_OVERFLOW = 1          # ibl_reg_ctx_status bits (include/ibloc.h)


def intensity_from_colors(colors) -> np.ndarray:
    """(r + g + b) / 3 in float64, rounded to fp32 -- the only colour quantity coloured ICP reads."""
    c = np.asarray(colors, dtype=np.float64)
    return ((c[:, 0] + c[:, 1] + c[:, 2]) / 3.0).astype(np.float32)


class MemoryShard:
    """Device-resident object memory: the stored embeddings (L2-normalised once), per-instance point clouds
    (x, y, z, intensity) and the spatial hash used by evaluate_transform.

    clouds=None: an embedding-only memory (BASELINE configs[3]: embed + match + assign, nothing to register against).
    shard=(rank, world): this rank keeps the embeddings of the instances [lo, hi) = parallel.shard_range(M, rank, world) only;
    instance indices everywhere else remain global.  The clouds stay replicated unless shard_clouds=True: then this rank also keeps
    only the clouds, cached features and evaluation grid of [lo, hi) (`clouds` / `colors` / `intensities` may be the full lists or just
    that range), registration jobs are routed between the ranks and the evaluation is reduced over them (routing.py)."""

    def __init__(self, ctx: RegContext, embeddings, clouds=None, colors=None, intensities=None, eval_threshold=0.02, device="cuda",
                 shard=None, shard_clouds=False, compact_features=False):
        self.ctx = ctx
        self.compact_features = bool(compact_features)     # resident instance features without their fp16 operand rows (168 B / point)
        self.device = torch.device(device)
        self.M = len(embeddings)
        self.shard = shard
        self.lo, self.hi = shard_range(self.M, *shard) if shard is not None else (0, self.M)
        own = embeddings[self.lo:self.hi]
        counts = [len(e) for e in own]
        self.emb_offsets_host = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        dim = np.asarray(embeddings[0]).shape[-1] if self.M else 0
        raw = np.concatenate([np.asarray(e, dtype=np.float32) for e in own]) if len(own) else np.zeros((0, dim), np.float32)
        if len(raw):
            self.mem_emb = match.normalize_rows(torch.from_numpy(np.ascontiguousarray(raw)).to(self.device))     # object_memory.py:922
        else:
            self.mem_emb = torch.zeros((0, dim), dtype=torch.float32, device=self.device)
        self.emb_offsets = torch.from_numpy(self.emb_offsets_host).to(self.device)
        self.eval_threshold = eval_threshold
        self._features = {}
        self.clouds = self.grid = None
        self.shard_clouds = bool(shard_clouds and shard is not None and clouds is not None)
        self.inst_sizes_all = None        # (M,) points per instance, gathered by the engine when the clouds are sharded
        if self.shard_clouds:
            def own_part(x):
                return x if x is None or len(x) == self.hi - self.lo else x[self.lo:self.hi]
            if clouds is not None and len(clouds) == self.M:
                self.inst_sizes_all = np.array([len(c) for c in clouds], dtype=np.int64)
            clouds, colors, intensities = own_part(clouds), own_part(colors), own_part(intensities)
        if clouds is not None:
            if intensities is None:
                intensities = [intensity_from_colors(c) for c in colors] if colors is not None else None
            self.clouds = CloudBatch.from_numpy(clouds, intensities, device=device)
            self.grid = MemGrid(ctx, self.clouds.pts4, cell=2 * eval_threshold)

    def features(self, voxel_size, local_dist_factor):
        """Normals, FPFH and colour gradients of every memory instance for these registration parameters: computed on first
        use, then resident in HBM (1.3 GB per 1 000 instances of 5 000 points; 0.84 GB with compact_features) for every later query."""
        key = (float(voxel_size), float(local_dist_factor))
        if key not in self._features:
            self._features[key] = instance_features_batch(self.ctx, self.clouds, voxel_size, 2.0 * (voxel_size * local_dist_factor),
                                                          compact=self.compact_features)
        return self._features[key]

    def close(self):
        """Frees the host side of the spatial hash (the device side lives in the context arena)."""
        if self.grid is not None:
            self.grid.close()
            self.grid = None
        self._features = {}


@dataclass
class FrameResult:
    pose: np.ndarray                      # reference-faithful [x, y, z, qx, qy, qz, qw] (object_memory.py:1124-1131)
    pose_corrected: np.ndarray            # same, with the best assignment's own means (App. B item 4 fixed)
    assignments: list = field(default_factory=list)
    records: list = field(default_factory=list)
    best: int = -1
    n_clean: int = 0                      # detected points of the frame left after the radius-outlier removal (:992-998)


@dataclass
class _Lane:
    ctx: RegContext                       # scratch arena of this lane's registration calls
    lane: int                             # encoder workspace index
    stream: "torch.cuda.Stream"


class LocaliseEngine:
    def __init__(self, memory: MemoryShard, encoder=None, assign_threads=0, k_hi=K_HI, k_lo=K_LO, group=None, rows_cap=224, comm=None):
        self.memory = memory
        self.encoder = encoder
        self.ctx = memory.ctx
        self.assign_threads = assign_threads or min(os.cpu_count() or 1, 16)
        self.k_hi, self.k_lo = int(k_hi), int(k_lo)
        self.reuse_features = True      # False: every assignment recomputes its features (same results, the reference's schedule)
        self.use_candidates = True      # False: the assignment search always runs on the full rows (what round 1 did)
        self.fused_stage_b = True       # False: stage B as a sequence of library calls issued from Python (what rounds 1-2 did)
        self._pool_a = None             # stage-A worker of localise_stream
        self._pool_w = None             # lane threads of localise_concurrent
        self._lanes = []
        # sharded memory: the stage-A collectives go through torch.distributed's group, or through the library's RCCL communicator
        # (parallel.RcclComm) when one is given; a one-rank `comm` exercises the exchange path on a single GPU
        self.exchange = ShardExchange(group, rows_cap, comm) if memory.shard is not None and (memory.shard[1] > 1 or comm is not None) else None
        self.stats = {"frames": 0, "fallback_frames": 0}        # how often the candidate search had to be redone on full rows
        self.route = None
        if memory.shard_clouds:                                 # collective: every rank constructs its engine
            from .routing import Transport
            self.route = Transport(group)
            if memory.inst_sizes_all is None:
                parts = self.route.all_gather_object(np.diff(memory.clouds.seg_off_host).astype(np.int64))
                memory.inst_sizes_all = np.concatenate(parts)
            assert len(memory.inst_sizes_all) == memory.M
            self.route_stats = {}

    def close(self):
        """Releases the lane arenas (multi-GB hipMalloc each) and worker threads now instead of at garbage collection."""
        for lane in self._lanes:
            lane.ctx.close()
        self._lanes = []
        for p in (self._pool_a, self._pool_w):
            if p is not None:
                p.shutdown(wait=True)
        self._pool_a = self._pool_w = None

    def check_status(self):
        """Reads (and clears) the registration context's device status word; warns when a scratch grid overflowed its cell budget
        (the neighbourhood search then took the slow path: results unaffected, time is)."""
        st = self.ctx.status()
        if st & IBL_ST_GRID_OVERFLOW:
            warnings.warn("registration scratch grid overflowed its cell budget: a neighbourhood search fell back to the slow path "
                          "(results unaffected); clouds this large want a larger arena or a coarser fpfh_voxel_size")
        return st

    # ---- stage A ---------------------------------------------------------------------------------------------------------
    def match_assign(self, q_per_frame, crops=None, det_emb=None, num_per_length=4, lane=0, tick=lambda name: None):
        """Stage A: embed + match + candidate selection on the current stream, assignment search on the host.  Returns the list
        (one entry per frame) of assignment lists `[[det_idx, mem_idx], ...]` with GLOBAL memory indices -- what the reference's
        SimVolume(closest_similarities).get_top_indices_from_subvolumes(4) returns (object_memory.py:974-982)."""
        mem = self.memory
        q_per_frame = np.asarray(q_per_frame, dtype=np.int32)
        F = len(q_per_frame)
        ex = self.exchange
        # input checks first; with a sharded memory the ranks then AGREE on the outcome before the step's collectives (one all-reduce
        # of flags): a rank that raised alone would leave the others waiting in the all-gather forever
        q_emb = np.minimum(q_per_frame, mem.M).astype(np.int32)      # Q > M truncates the detections that are matched (:918-920)
        err = None
        if (q_per_frame > 7).any():
            err = ValueError("more than 7 detections in a frame: select the 7 largest first (object_memory.py:900-908)")
        elif (q_emb < np.minimum(q_per_frame, 3)).any():
            err = AssertionError("fewer memory objects than the sub-volume dimension (reference asserts at similarity_volume.py:112)")
        elif det_emb is None and self.encoder is None:
            err = ValueError("no encoder: pass det_emb")
        elif ex is not None and int(q_per_frame.sum()) > ex.cap:
            err = ValueError(f"{int(q_per_frame.sum())} query rows in a step, rows_cap is {ex.cap}")
        if ex is not None:
            any_err, any_done, _ = ex.agree(err is not None, False, mem.device)
            if err is None and any_done:
                err = RuntimeError("sharded memory: another rank has run out of batches (every rank must localise the same number of batches)")
            if err is None and any_err:
                err = RuntimeError("sharded memory: another rank rejected its inputs for this step; no rank runs it")
        if err is not None:
            raise err
        row0 = np.concatenate([[0], np.cumsum(q_per_frame)]).astype(np.int32)
        if det_emb is None:
            det_emb = self.encoder.embed(crops, lane=lane)
        else:
            det_emb = torch.as_tensor(det_emb, dtype=torch.float32, device=mem.device).contiguous()
        tick("embed")
        detn = match.normalize_rows(det_emb)                                        # :924
        R = detn.shape[0]
        if ex is None:
            val, idx, cnt, aug = match.match_topk(detn, mem.mem_emb, mem.emb_offsets, self.k_hi, self.k_lo, 0)   # :933-936 + sim_volume :13-18
            tick("match")
            if not self.use_candidates:
                aug_h = aug.cpu().numpy()
                tick("d2h")
                return self._assign_full(aug_h, row0, q_emb, num_per_length, tick)
            val_h, idx_h = val.cpu().numpy(), idx.cpu().numpy()
            cnt_h = cnt.cpu().numpy().sum(axis=1).astype(np.int32)
        else:
            val_h, idx_h, cnt_h, aug = sharded_candidates(
                ex, detn, lambda rows: match.match_topk(rows, mem.mem_emb, mem.emb_offsets, self.k_hi, self.k_lo, mem.lo))
            tick("match")
        tick("d2h")
        assns, exact = assign_candidates(val_h, idx_h, cnt_h, row0[:-1], q_emb, mem.M, self.k_hi, self.k_lo, num_per_length,
                                         self.assign_threads)
        redo = np.nonzero(~exact)[0]
        self.stats["frames"] += F
        self.stats["fallback_frames"] += len(redo)
        if ex is not None:
            if ex.any_flag(len(redo) > 0, detn.device):                             # rare: some rank could not prove a frame
                full = ex.gather_blocks(aug, mem.M)[:R]
                if len(redo):
                    self._redo_full(full, redo, row0, q_emb, num_per_length, assns)
        elif len(redo):
            self._redo_full(aug, redo, row0, q_emb, num_per_length, assns)
        tick("assign")
        return assns

    def _redo_full(self, aug_dev, frames, row0, q_emb, num_per_length, assns):
        """the exact search on the full rows of the frames whose candidate search could not be proved (ties at the threshold)"""
        rows = np.concatenate([np.arange(row0[f], row0[f] + q_emb[f]) for f in frames])
        sub = aug_dev[torch.from_numpy(rows).to(aug_dev.device)].cpu().numpy()
        M = aug_dev.shape[1] - 1
        aug_f = np.ones((len(frames), 7, M + 1), dtype=np.float16)
        o = 0
        for i, f in enumerate(frames):
            aug_f[i, :q_emb[f]] = sub[o:o + q_emb[f]]
            o += q_emb[f]
        res = assign_batch(aug_f, q_emb[frames], num_per_length, self.assign_threads)
        for i, f in enumerate(frames):
            assns[f] = res[i]

    def _assign_full(self, aug_h, row0, q_emb, num_per_length, tick):
        F = len(q_emb)
        aug_f = np.ones((F, 7, aug_h.shape[1]), dtype=np.float16)
        for f in range(F):
            aug_f[f, :q_emb[f]] = aug_h[row0[f]:row0[f] + q_emb[f]]
        assns = assign_batch(aug_f, q_emb, num_per_length, self.assign_threads)
        tick("assign")
        return assns

    # ---- schedulers --------------------------------------------------------------------------------------------------------
    def localise_concurrent(self, batches, workers=2, worker_arena_bytes=8 << 30, **kw):
        """`localise_batch` for a sequence of frame batches (dicts like `localise_stream` takes) on `workers` concurrent lanes: every
        lane is a host thread with its own HIP stream, registration scratch arena and encoder workspace, and runs whole batches
        start to end, so the host phases and launch gaps of one batch are filled by the kernels of the others.  The memory (its
        embeddings, clouds, resident instance features and spatial hash) is shared read-only.  Results are those of
        `localise_batch`, yielded in order."""
        import queue
        from collections import deque
        if self.exchange is not None:
            raise ValueError("localise_concurrent with a sharded memory: collectives must be issued by one thread (use localise_stream)")
        dev = self.memory.mem_emb.device
        if self.reuse_features and self.memory.clouds is not None:     # build the resident memory features before the lanes start
            self.memory.features(kw.get("fpfh_voxel_size", 0.05), kw.get("fpfh_local_dist_factor", 0.4))
        while len(self._lanes) < workers:
            k = len(self._lanes)
            self._lanes.append(_Lane(RegContext(worker_arena_bytes), k, torch.cuda.Stream(device=dev)))
        free = queue.Queue()
        cur = torch.cuda.current_stream(dev)
        for lane in self._lanes[:workers]:
            lane.stream.wait_stream(cur)                     # inputs (crops, clouds, memory) were produced on the caller's stream
            free.put(lane)

        def run(b):
            lane = free.get()
            try:
                torch.cuda.set_device(dev)
                args = dict(kw)
                for k in ("seed", "job_id_base"):
                    if k in b:
                        args[k] = b[k]
                with torch.cuda.stream(lane.stream):
                    res = self.localise_batch(b["det"], b["q_per_frame"], crops=b.get("crops"), det_emb=b.get("det_emb"), _slot=lane, **args)
                    lane.stream.synchronize()
                return res
            finally:
                free.put(lane)

        if self._pool_w is None or self._pool_w._max_workers < workers:
            self._pool_w = ThreadPoolExecutor(max_workers=workers)
        pending = deque()
        for b in batches:
            pending.append(self._pool_w.submit(run, b))
            if len(pending) > workers:                       # keep the lanes fed, bound what is in flight
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()

    def localise_stream(self, batches, **kw):
        """Pipelined form of `localise_batch` for a sequence of frame batches (dicts with det, q_per_frame and crops or det_emb,
        optionally seed / job_id_base): stage A (embed, match, candidates, assignment search, every collective) of batch k + 1
        runs on a second stream, driven by a worker thread, while stage B of batch k (registration) executes; results are those
        of `localise_batch`, yielded in order.  The two stages touch disjoint state: the encoder / match workspaces belong to
        stage A, the registration arena to stage B."""
        depth = max(1, int(kw.pop("lookahead", os.environ.get("IBL_STAGE_A_LOOKAHEAD", "2"))))
        if self.route is not None:
            # sharded clouds: stage B has collectives of its own (routing.py); two threads issuing collectives on one group could pair
            # them up differently on different ranks, so the batches run one after the other
            for b in batches:
                args = {k: b[k] for k in ("det", "q_per_frame", "crops", "det_emb", "seed", "job_id_base") if k in b}
                yield self.localise_batch(**args, **kw)
            return
        dev = self.memory.mem_emb.device
        side = torch.cuda.Stream(device=dev)      # equal priority: raising either stage's stream priority measured 7-8 % slower
        main = torch.cuda.current_stream(dev)
        npl = kw.get("num_per_length", 4)

        def stage_a(b, ready):
            torch.cuda.set_device(dev)                      # worker thread: current device and stream are per thread
            with torch.cuda.stream(side):
                side.wait_event(ready)                      # this batch's inputs (crops) were produced on the caller's stream
                return self.match_assign(b["q_per_frame"], b.get("crops"), b.get("det_emb"), npl)

        def submit(b):
            # `batches` may be a lazy iterator whose crops are written by kernels on the caller's stream as it is advanced: the
            # event is recorded when the batch is pulled, and stage A waits for it before touching the crops
            ready = torch.cuda.Event()
            ready.record(main)
            return self._pool_a.submit(stage_a, b, ready)

        def stage_done():
            # sharded memory: tell the other ranks this one has no batch left -- if any of them still has one, every rank raises
            # instead of that rank waiting for this one in its all-gather (issued by the stage-A thread, like every collective)
            torch.cuda.set_device(dev)
            with torch.cuda.stream(side):
                _, _, any_active = self.exchange.agree(False, True, dev)
            if any_active:
                raise RuntimeError("sharded memory: the ranks were given different numbers of batches")

        # Stage A runs up to `lookahead` batches ahead of stage B (default 2, $IBL_STAGE_A_LOOKAHEAD).  With one batch of lookahead stage A of
        # batch k + 1 starts together with stage B of batch k and ends first (17 against 21 ms of GPU work on the bench workload), so the
        # end of every stage B -- the late ICP iterations, launches of a few workgroups each -- ran with most of the chip idle; a second
        # batch in flight keeps the encoder's GEMMs queued behind them.  The worker thread is the only one that issues stage-A work (and
        # every collective), in batch order, whatever the depth.
        fut_done = None
        it = iter(batches)
        pending = deque()
        exhausted = False
        if self._pool_a is None:
            self._pool_a = ThreadPoolExecutor(max_workers=1)

        def fill():
            nonlocal exhausted, fut_done
            while not exhausted and len(pending) < depth:
                b = next(it, None)
                if b is None:
                    exhausted = True
                    if self.exchange is not None:
                        fut_done = self._pool_a.submit(stage_done)
                else:
                    pending.append((b, submit(b)))

        try:
            fill()
            while pending:
                cur, fut = pending[0]
                assns = fut.result()
                pending.popleft()
                fill()
                args = dict(kw)
                for k in ("seed", "job_id_base"):
                    if k in cur:
                        args[k] = cur[k]
                yield self.localise_batch(cur["det"], cur["q_per_frame"], assns=assns, **args)
            if fut_done is not None:
                fut_done.result()
        finally:
            # an abandoned generator or a raising batch must not leave stage A running on the side stream with the lane-0 encoder
            # workspace: a later localise_batch on the caller's stream would race with it
            for _, f in pending:
                try:
                    f.result()
                except Exception:
                    pass
            side.synchronize()

    # ---- sharded clouds: routed registration + reduced evaluation (routing.py) ------------------------------------------------
    def _register_routed(self, ctx, clean, det_feat, mem_feat, job_src, job_tgt, job_id_base, voxel, gdf, ldf, seed, ransac_max_iter):
        from .routing import InstanceStore, routed_register
        mem = self.memory
        assert det_feat is not None and mem_feat is not None, "sharded clouds need reuse_features (the cached instance features travel)"
        arrays = {"pts": mem.clouds.pts4, "normals": mem_feat.normals, "fpfh": mem_feat.fpfh, "fpfh_norm": mem_feat.fpfh_norm}
        if mem_feat.fpfh_split is not None:                 # (compact features travel without their operand rows)
            arrays["fpfh_split"] = mem_feat.fpfh_split
        if mem_feat.grad is not None:
            arrays["grad"] = mem_feat.grad
        store = InstanceStore(mem.lo, mem.clouds.seg_off_host, arrays, {"bbox": mem_feat.bbox})

        def compute(pool_pts, pool_off, n_home, arr, per_inst, mem_off, js, jt, ids):
            det_pool = CloudBatch(pool_pts.contiguous(), pool_off.astype(np.int32))
            feat = det_feat
            if det_pool.n_seg > n_home:                      # detected segments shipped here: their features, computed where they run
                b0 = int(pool_off[n_home])
                extra = instance_features_batch(ctx, CloudBatch(pool_pts[b0:].contiguous(), (pool_off[n_home:] - b0).astype(np.int32)), voxel)
                feat = InstanceFeatures(torch.cat([det_feat.normals[:clean.n], extra.normals]), torch.cat([det_feat.fpfh[:clean.n], extra.fpfh]),
                                        torch.cat([det_feat.fpfh_split[:clean.n], extra.fpfh_split]),
                                        torch.cat([det_feat.fpfh_norm[:clean.n], extra.fpfh_norm]), None,
                                        np.concatenate([det_feat.bbox[:clean.n_seg], extra.bbox]), voxel, 0.0)
            mem_pool = CloudBatch(arr["pts"].contiguous(), np.asarray(mem_off).astype(np.int32))
            mf = InstanceFeatures(arr["normals"].contiguous(), arr["fpfh"].contiguous(), arr["fpfh_split"].contiguous() if "fpfh_split" in arr else None,
                                  arr["fpfh_norm"].contiguous(), arr["grad"].contiguous() if "grad" in arr else None,
                                  np.ascontiguousarray(per_inst["bbox"]), mem_feat.voxel_size, mem_feat.grad_radius)
            r = register_batch(ctx, det_pool, mem_pool, js, jt, voxel, gdf, ldf, seed=seed, ransac_max_iter=ransac_max_iter,
                               have_colors=True, center=True, det_features=feat, mem_features=mf, job_ids=ids)
            return {k: r[k] for k in ("T", "rmse", "fitness", "means", "T_ransac", "ransac_stats")}

        J = len(job_src)
        ids = (np.uint32(job_id_base) + np.arange(J, dtype=np.uint32)).astype(np.uint32)
        src = np.asarray(job_src, dtype=np.int64).reshape(-1, 3)
        tgt = np.asarray(job_tgt, dtype=np.int64).reshape(-1, 3)
        params = (int(seed), float(voxel), float(gdf), float(ldf), int(ransac_max_iter))     # a shipped job runs with these: equal on all ranks
        reg, _ = routed_register(self.route, mem.M, clean.pts4, clean.seg_off_host, src, tgt, ids, store, mem.inst_sizes_all, compute,
                                 self.route_stats, params=params)
        return reg

    def _evaluate_routed(self, ctx, clean, jb, je, G, thr):
        from .routing import routed_evaluate
        mem = self.memory
        thr = thr if thr is not None else mem.eval_threshold

        def local(pts, b, e, T):
            return evaluate_points(ctx, mem.grid, pts.contiguous(), b, e, T, thr)[0]

        return routed_evaluate(self.route, clean.pts4, jb, je, G, local)

    # ---- one batch -----------------------------------------------------------------------------------------------------------
    def localise_batch(self, det: CloudBatch, q_per_frame, crops=None, det_emb=None, fpfh_voxel_size=0.05,
                       fpfh_global_dist_factor=2, fpfh_local_dist_factor=0.4, outlier_radius=0.05, outlier_nb_points=8,
                       seed=0, job_id_base=0, ransac_max_iter=4000000, num_per_length=4, eval_threshold=None, timings=None,
                       assns=None, _slot=None, register=True, ransac_fixed_budget=False):
        """det: detected clouds of all frames (segments in frame order, <= 7 per frame); crops: uint8 tensor
        (sum Q, H, W, 3) or list of arrays (DATOR: the (rgb, depth) pair), or det_emb: (sum Q, D) precomputed embeddings (or
        assns: the assignment lists stage A of `localise_stream` produced).  register=False stops after the assignment search
        (embedding-only memories): the results carry the assignment lists, `best` = 0 and identity poses; `det` may be None."""
        mem = self.memory
        ctx, lane = (self.ctx, 0) if _slot is None else (_slot.ctx, _slot.lane)
        q_per_frame = np.asarray(q_per_frame, dtype=np.int32)
        F = len(q_per_frame)
        row0 = np.concatenate([[0], np.cumsum(q_per_frame)]).astype(np.int64)
        register = register and mem.clouds is not None
        assert not register or det.n_seg == int(row0[-1])
        ev = []

        def tick(name):
            if timings is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))

        def close_timings():
            if timings is not None:
                torch.cuda.synchronize()
                for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
                    timings[n1] = timings.get(n1, 0.0) + e0.elapsed_time(e1)

        tick("start")
        # ---- stage A: embed + match + assign (:911-982) -----------------------------------------------
        if assns is None:
            assns = self.match_assign(q_per_frame, crops, det_emb, num_per_length, lane, tick)
        results = [FrameResult(np.array([0., 0., 0., 0., 0., 0., 1.]), np.array([0., 0., 0., 0., 0., 0., 1.])) for _ in range(F)]
        if not register:
            for f in range(F):
                results[f].assignments = assns[f]
                results[f].best = 0 if assns[f] else -1
            close_timings()
            return results
        # ---- stage B in one library call (ibl_register_evaluate_batch, csrc/localise.hip): what the staged sequence below computes, bit
        # for bit (tests/test_gpu_engine.py), without Python or torch between the launches.  The staged form remains for
        # reuse_features = False and for sharded clouds (whose stages are separated by collectives).
        if self.fused_stage_b and self.route is None and self.reuse_features:
            if not any(len(a) for a in assns):
                close_timings()
                return results
            thr = eval_threshold if eval_threshold is not None else mem.eval_threshold
            r = register_evaluate_batch(ctx, det, q_per_frame, assns, mem.clouds, mem.features(fpfh_voxel_size, fpfh_local_dist_factor), mem.grid,
                                        fpfh_voxel_size, fpfh_global_dist_factor, fpfh_local_dist_factor, outlier_radius, outlier_nb_points, thr,
                                        seed=seed, job_id_base=job_id_base, ransac_max_iter=ransac_max_iter, fixed_budget=ransac_fixed_budget)
            if timings is not None:
                timings["reuse"] = r["reuse"].tolist()
            tick("stage_b")              # (the split of stage B into its stages: the library's in-process stage timer, prof.stage_rooflines)
            self._assemble(results, assns, row0, r["clean_off"], r, r["T_global"], r["full_rmse"], r["full_fitness"])
            tick("select")
            close_timings()
            return results
        # ---- clean the detected clouds (:992-998) ---------------------------------------------------
        keep = radius_outlier_batch(ctx, det, outlier_radius, outlier_nb_points)
        keepb = keep.bool()
        csum = torch.cumsum(keep.to(torch.int32), 0)
        csum0 = torch.cat([torch.zeros(1, dtype=torch.int32, device=csum.device), csum])
        new_off = csum0[det.seg_off.long()]
        clean_pts = det.pts4[keepb].contiguous()
        new_off_h = new_off.cpu().numpy().astype(np.int32)
        clean = CloudBatch(clean_pts, new_off_h)
        tick("outlier")
        # ---- registration jobs (:1020-1106) ----------------------------------------------------------
        job_frame, job_src, job_tgt = [], [], []
        for f in range(F):
            for a in assns[f]:
                job_frame.append(f)
                job_src.append([int(row0[f]) + d for d, m in a] + [-1] * (3 - len(a)))
                job_tgt.append([m for d, m in a] + [-1] * (3 - len(a)))
        if not job_frame and self.route is None:        # (sharded clouds: the collectives below still need this rank)
            close_timings()
            return results
        # instance features (normals / FPFH, once per cloud instead of once per assignment): the memory's are resident (built on
        # first use), the detections' are computed here once per frame batch
        det_feat = mem_feat = None
        if self.reuse_features:
            det_feat = instance_features_batch(ctx, clean, fpfh_voxel_size)
            mem_feat = mem.features(fpfh_voxel_size, fpfh_local_dist_factor)
        tick("det_features")
        if self.route is not None:
            reg = self._register_routed(ctx, clean, det_feat, mem_feat, job_src, job_tgt, job_id_base, fpfh_voxel_size,
                                        fpfh_global_dist_factor, fpfh_local_dist_factor, seed, ransac_max_iter)
            if not job_frame:
                self._evaluate_routed(ctx, clean, [], [], np.zeros((0, 4, 4)), eval_threshold)
                close_timings()
                return results
        else:
            reg = register_batch(ctx, clean, mem.clouds, job_src, job_tgt, fpfh_voxel_size, fpfh_global_dist_factor,
                                 fpfh_local_dist_factor, seed=seed, job_id_base=job_id_base, ransac_max_iter=ransac_max_iter,
                                 have_colors=True, center=True, det_features=det_feat, mem_features=mem_feat, fixed_budget=ransac_fixed_budget)
            if timings is not None:
                timings["reuse"] = reg["reuse"].tolist()
        tick("register")
        # global-frame transforms (:1096-1101)
        T = reg["T"]
        means = reg["means"]
        G = T.copy()
        Rm = T[:, :3, :3]
        dm = means[:, 0]
        G[:, :3, 3] = T[:, :3, 3] + means[:, 1] - ((Rm[:, :, 0] * dm[:, 0:1] + Rm[:, :, 1] * dm[:, 1:2]) + Rm[:, :, 2] * dm[:, 2:3])
        jb = [int(new_off_h[row0[f]]) for f in job_frame]
        je = [int(new_off_h[row0[f + 1]]) for f in job_frame]
        thr = eval_threshold if eval_threshold is not None else mem.eval_threshold
        if self.route is not None:
            full_fit, full_rmse = self._evaluate_routed(ctx, clean, jb, je, G, thr)
        else:
            full_rmse, full_fit = evaluate_batch(ctx, mem.grid, clean.pts4, jb, je, G, thr)     # :1104
        tick("evaluate")
        self._assemble(results, assns, row0, new_off_h, reg, G, full_rmse, full_fit)
        tick("select")
        close_timings()
        return results

    @staticmethod
    def _assemble(results, assns, row0, new_off_h, reg, G, full_rmse, full_fit):
        """selection + pose (:1111-1131) from the per-job arrays (frames in order, a frame's jobs in assignment order)"""
        T, means = reg["T"], reg["means"]
        j = 0
        picked = []                          # (frame, first job, best job) of the frames that have assignments
        for f in range(len(assns)):
            n_a = len(assns[f])
            if n_a == 0:
                continue
            ff = full_fit[j:j + n_a]
            best = max(range(n_a), key=lambda k: (ff[k], -k))      # first maximum == sorted(..., reverse=True)[0], stable (:1111)
            picked.append((f, j, best))
            j += n_a
        if picked:
            quats = Rotation.from_matrix(np.stack([T[j0 + b, :3, :3] for _, j0, b in picked])).as_quat()
        for (f, j0, best), q in zip(picked, quats if picked else []):
            n_a = len(assns[f])
            recs = [dict(assn=assns[f][k], T=T[j0 + k], rmse=reg["rmse"][j0 + k], fitness=reg["fitness"][j0 + k],
                         full_rmse=full_rmse[j0 + k], full_fitness=full_fit[j0 + k], T_global=G[j0 + k],
                         detected_mean=means[j0 + k, 0], memory_mean=means[j0 + k, 1], ransac_stats=reg["ransac_stats"][j0 + k])
                    for k in range(n_a)]
            R = T[j0 + best, :3, :3]
            t = T[j0 + best, :3, 3]
            last = j0 + n_a - 1
            t_ref = t + means[last, 1] - R @ means[last, 0]                       # stale means of the LAST assignment (:1127)
            t_fix = t + means[j0 + best, 1] - R @ means[j0 + best, 0]
            results[f] = FrameResult(np.concatenate((t_ref, q)), np.concatenate((t_fix, q)), assns[f], recs, best,
                                     int(new_off_h[row0[f + 1]] - new_off_h[row0[f]]))
        return results
