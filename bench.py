#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the OpenPose BODY_25 bottom-up path at
368x656 (preprocess + VGG-19/6-stage PAF network + full PAF decode) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run);
frames shard independently across ranks, no collective in the compute path;
at N > 1 one RCCL gather (dst = rank 0) of the packed, padded keypoint record
per step, at N = 1 no collective at all.  A step is one batch of 32 synthetic
frames per GPU, already resident in HBM:
  u8 frames -> ph_preprocess -> ph_forward (MFMA implicit-GEMM convs) ->
  ph_decode_v2 (upsample+NMS, PAF matching, assembly) -> keypoints to host.
Weights are He-normal random (no model files exist offline): same graph, same
FLOPs, `data: synthetic`.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pose-estimation_amd"))
sys.path.insert(0, ROOT)

NET_H, NET_W, BATCH = 368, 656, 32
MFMA_PEAK_TFLOPS = 2500.0      # MI355X bf16 dense (MI355X_MICROARCH.md)
MAX_PEOPLE = 128


def cpu_baseline(net, weights, model, n_frames, seed):
    """The oracle (torch fp32 forward + NumPy V2 decode) timed on the host cores:
    a CPU stand-in for the reference's cv2.dnn / pyopenpose path (cv2, Caffe and
    OpenPose are not installed anywhere this runs)."""
    import torch
    from oracle import decode_v2, net_ref
    from posehip import synth
    # threads actually used: the box's CPU share (OMP_NUM_THREADS, 16 per GPU on the pool's boxes -- os.cpu_count()
    # there shows the whole machine, which one GPU job may not use) or, where that is not set, every host core
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    frames = synth.synth_frames(seed, n_frames, NET_H, NET_W)
    t0 = time.time()
    for i in range(n_frames):
        x = frames[i:i + 1].astype(np.float32).transpose(0, 3, 1, 2) * np.float32(1 / 256.0) - np.float32(0.5)
        out = net_ref.forward(net, weights, x)
        decode_v2.decode(out[0], model, NET_H, NET_W, 1.0)
    dt = time.time() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": cores, "host_cpu_count": os.cpu_count(),
            "kind": "port",
            "sample": f"{n_frames} synthetic 368x656 frames, batch 1, torch-CPU fp32 forward + NumPy V2 decode "
                      f"(oracle/net_ref.py + oracle/decode_v2.py); cv2.dnn/pyopenpose unavailable"}


VALU_LANE_OPS_PER_S = 256 * 64 * 2.4e9    # 256 CUs x 64 lanes x 2.4 GHz: one fp32 VALU operation per lane and clock
# VALU instructions the NMS kernel issues per upsampled pixel, measured: SQ_INSTS_VALU x 64 lanes / pixels
# (profiles/r03_pmc_nms_insts.txt: 88.2 M wave instructions per 32-frame launch; a packed fp32 instruction counts once)
NMS_OPS_PER_PIXEL = 29


def tail_block(B, src_h, src_w, pre_ms, decode_ms, paf_pairs):
    """The tail beside the convs, per launch of B frames; durations are HIP events of this run.
      preprocess    HBM-bound: read H0*W0*3 u8 + write Hn*Wn*3 bf16 per frame (the buffer pads to 8 channels)
      upsample+NMS  VALU-bound: the fused kernel reads 25 source planes once (`hbm_bytes`) and evaluates the bicubic
                    + 3x3 test for every upsampled pixel in registers: NMS_OPS_PER_PIXEL fp32 operations per pixel
                    against the chip's fp32 issue rate.  `unfused_algorithmic_bytes` is what upstream's
                    resizeAndMerge -> nms moves (100.6 MB/frame) -- context, never a roofline fraction.
      PAF scoring   VALU + LDS gather: pairs x <=25 samples x 2 channels x 4 B gathered from LDS-resident planes
      assembly      latency-bound (a block per frame)"""
    hbm = 8000.0
    hs, ws = NET_H // 8, NET_W // 8
    out = {}
    pre_bytes = B * (src_h * src_w * 3 + NET_H * NET_W * 3 * 2)
    gbs = pre_bytes / (pre_ms * 1e-3) / 1e9 if pre_ms > 0 else 0.0
    out["preprocess_kernel"] = {"bound": "hbm", "algorithmic_bytes": int(pre_bytes), "us": round(pre_ms * 1e3, 1),
                                "GB_per_s": round(gbs, 1), "frac_of_8TBps": round(gbs / hbm, 4)}
    nms_ms = float(decode_ms[0])
    ops = B * 25 * NET_H * NET_W * NMS_OPS_PER_PIXEL
    rate = ops / (nms_ms * 1e-3) if nms_ms > 0 else 0.0
    out["nms_detect_cols_kernel+nms_refine_kernel"] = {
        "bound": "valu", "us": round(nms_ms * 1e3, 1), "upsampled_pixels": B * 25 * NET_H * NET_W,
        "ops_per_pixel": NMS_OPS_PER_PIXEL, "achieved_Tops_per_s": round(rate / 1e12, 2),
        "peak_Tops_per_s": round(VALU_LANE_OPS_PER_S / 1e12, 2), "frac_of_fp32_issue_rate": round(rate / VALU_LANE_OPS_PER_S, 4),
        "hbm_bytes": int(B * 25 * hs * ws * 4),
        "unfused_algorithmic_bytes": int(B * (78 * hs * ws * 4 + 78 * NET_H * NET_W * 4 + 25 * NET_H * NET_W * 4))}
    paf_ms = float(decode_ms[1])
    out["paf_score_kernel"] = {"bound": "valu+lds", "us": round(paf_ms * 1e3, 1), "candidate_pairs": int(paf_pairs),
                               "gathered_bytes": int(paf_pairs * 25 * 2 * 4)}
    out["assemble_merge_kernel"] = {"bound": "latency", "us": round(float(decode_ms[2]) * 1e3, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 steps = 1.8 s: with three batches in flight a 20-step region (0.19 s) is 3 % pipeline fill and drain
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--per-op", action="store_true", help="print a per-op time table to stderr")
    ap.add_argument("--streams", type=int, default=3, help="batches in flight per GPU (engines / HIP streams)")
    ap.add_argument("--frame-size", type=str, default=f"{NET_H}x{NET_W}",
                    help="HxW of the source frames (1080x1920 = BASELINE configs[4]: 1080p stream, '-1x368' "
                         "net resolution, keypoints scaled back to source pixels); default = configs[1]")
    ap.add_argument("--h2d", action="store_true",
                    help="frames start in pinned HOST memory and are copied in every step (PCIe-inclusive "
                         "rate for DESIGN.md; never the headline value)")
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="when --steps is below 100: steps of a longer leg reported as `sustained` beside the headline "
                         "(0 = skip)")
    ap.add_argument("--cpu-frames", type=int, default=24,
                    help="frames of the CPU baseline sample (about 12 s of host work on 16 cores; 0 = skip)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 when its first communicator
    # comes up, so everything but the result line is sent to stderr at the descriptor level.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default): with three engine
    # streams, the copy stream of the host-frame leg and the default stream, the copy stream shares a queue with an
    # engine, and every 3.5 ms frame copy then holds up that engine's kernels (measured: 0.82 of the resident rate with
    # 4 queues, 0.997 with 8; 0.999 with 4 queues and two engines).  One queue per stream, set before HIP comes up.
    from posehip.pipeline import ensure_hw_queues
    hw_queues = ensure_hw_queues(max(1, args.streams) + 2)

    import torch
    import torch.distributed as dist
    from posehip import _ffi, caffe_io, graph
    from posehip.engine import Engine
    from posehip.tables import BODY_25

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # POSEHIP_BENCH_BACKEND=gloo + several ranks on one card rehearses the N > 1 control flow on a 1-GPU box
    backend = os.environ.get("POSEHIP_BENCH_BACKEND", "nccl")
    local_rank %= torch.cuda.device_count()
    # POSEHIP_BENCH_FORCE_GATHER=1 runs the RCCL gather path with a one-rank group (rehearsal on a 1-GPU box)
    gather = world > 1 or bool(os.environ.get("POSEHIP_BENCH_FORCE_GATHER"))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif gather:
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:29531", rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank) if backend == "nccl" else None)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.batch

    net = caffe_io.parse_prototxt(caffe_io.body25_prototxt(NET_H, NET_W))
    weights = graph.random_weights(net, seed=0)
    plan = graph.build_plan(net, weights)
    # `--streams S` batches in flight per GPU: S engines (own HIP stream, own activation buffers, shared
    # nothing), step i runs on engine i % S.  Default 3: measured +1.8 % over 2; 4 is slower (4 hardware queues).  Consecutive batches are independent, so the latency-bound
    # decode kernels and the partial last waves of one batch's convs overlap the next batch's convs.
    S = max(1, args.streams)
    tstreams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    engs = []
    for ts in tstreams:
        e = Engine(local_rank, ts.cuda_stream)
        e.build(plan, B)
        engs.append(e)
    eng = engs[0]
    flops_frame = plan.flops_per_frame()
    conv_mask = np.array([o["type"] == _ffi.PH_OP_CONV for o in plan.ops])

    from posehip import synth
    src_h, src_w = (int(v) for v in args.frame_size.lower().split("x"))
    from posehip.api_native import scale_net_to_output
    kp_scale = scale_net_to_output(src_w, src_h, NET_W, NET_H)   # OpenPose's mScaleNetToOutput (1.0 at 656x368)
    frames_host = torch.from_numpy(synth.synth_frames(1000 + rank, B, src_h, src_w)).pin_memory()
    J = BODY_25.n_parts
    bufs = []
    for k in range(S):
        d = {"frames": frames_host.to(dev),
             "n": torch.empty((B,), dtype=torch.int32).pin_memory(),
             "kp": torch.empty((B, MAX_PEOPLE, J, 3), dtype=torch.float32).pin_memory(),
             "sc": torch.empty((B, MAX_PEOPLE), dtype=torch.float32).pin_memory()}
        if gather:  # 39 KB/frame: latency-bound, far below one xGMI link
            # one packed record per rank and step (keypoints | scores | counts), so ONE collective per step
            n_kp, n_sc = B * MAX_PEOPLE * J * 3, B * MAX_PEOPLE
            d["pack"] = torch.empty((n_kp + n_sc + B,), dtype=torch.float32, device=dev)
            d["gkp"] = d["pack"][:n_kp].view(B, MAX_PEOPLE, J, 3)
            d["gsc"] = d["pack"][n_kp:n_kp + n_sc].view(B, MAX_PEOPLE)
            d["gn"] = d["pack"][n_kp + n_sc:].view(torch.int32)
        bufs.append(d)
    g = {}
    if gather:
        gdev = dev if backend == "nccl" else torch.device("cpu")
        # rank 0 only: one gathered record set per engine (gathers of different engines may overlap)
        for d in bufs:
            d["gbins"] = [torch.empty((d["pack"].numel(),), dtype=torch.float32, device=gdev)
                          for _ in range(world)] if rank == 0 else None
            d["hpack"] = torch.empty((d["pack"].numel(),), dtype=torch.float32).pin_memory()
    # Frames that arrive in HOST memory (a video stream: BASELINE configs[4]) are brought in by a dedicated copy
    # stream, one batch ahead: every engine owns two device frame buffers, the copy of its NEXT batch is issued right
    # after this batch's preprocess has been enqueued and is handed over by an event, so no copy sits on the critical
    # path of an engine's stream and no host-side wait is needed for it.
    copy_stream = torch.cuda.Stream(device=dev)
    for d in bufs:
        d["fr"] = [d["frames"], None]
        d["copied"] = [torch.cuda.Event(), torch.cuda.Event()]
        d["pre_done"] = [None, None]
        d["cnt"], d["prefetched"] = 0, False

    def issue_copy(d, j):
        if d["fr"][j] is None:
            d["fr"][j] = torch.empty_like(d["frames"])
        with torch.cuda.stream(copy_stream):
            if d["pre_done"][j] is not None:       # the preprocess that last read this buffer (two batches ago)
                copy_stream.wait_event(d["pre_done"][j])
            d["fr"][j].copy_(frames_host, non_blocking=True)
            d["copied"][j].record(copy_stream)

    trace = (lambda *a: print(f"[rank {rank}]", *a, file=sys.stderr, flush=True)) if os.environ.get("BENCH_TRACE") \
        else (lambda *a: None)
    trace("engines built")

    def step(i):
        """One pass of the hot path over one batch of B frames (enqueued; completes asynchronously)."""
        k = i % S
        e, ts, d = engs[k], tstreams[k], bufs[k]
        ts.synchronize()           # the batch this engine ran S steps ago has delivered its keypoints
        if i >= S and not os.environ.get("BENCH_DIAG_NO_DECODE"):
            e.decode_v2_status()   # raises if that batch's decode dropped people (person table overflow)
        if args.h2d:
            j = d["cnt"] % 2
            if not d["prefetched"]:
                issue_copy(d, j)
            ts.wait_event(d["copied"][j])
            e.preprocess(d["fr"][j], _ffi.PH_PRE_NATIVE)
            if d["pre_done"][j] is None:
                d["pre_done"][j] = torch.cuda.Event()
            d["pre_done"][j].record(ts)
            d["cnt"] += 1
            issue_copy(d, d["cnt"] % 2)            # this engine's next batch, S steps ahead of its use
            d["prefetched"] = True
        else:
            e.preprocess(d["frames"], _ffi.PH_PRE_NATIVE)
        e.forward(B)
        if not os.environ.get("BENCH_DIAG_NO_DECODE"):   # diagnostic only: what the decode costs end to end
            e.decode_v2_dev(BODY_25, B, NET_H, NET_W, kp_scale, MAX_PEOPLE)
            # results to pinned host memory (rank-local) ...
            e.decode_v2_results(d["n"], d["kp"], d["sc"])
        if gather and backend == "nccl":  # ... and ONE RCCL gather of the packed, padded record to rank 0
            e.decode_v2_results(d["gn"], d["gkp"], d["gsc"])
            # Issued with the ENGINE's stream current: the collective waits for this batch's decode only, and
            # only this engine's next batch waits for the collective.  (Issuing every gather from one common
            # stream and making the engines wait for that stream serializes the batches in flight: -12 %.)
            with torch.cuda.stream(ts):
                dist.gather(d["pack"], d["gbins"], dst=0)
        elif gather:  # rehearsal backend (gloo): same control flow, gather of a host copy of the record
            e.decode_v2_results(d["gn"], d["gkp"], d["gsc"])
            with torch.cuda.stream(ts):
                d["hpack"].copy_(d["pack"], non_blocking=True)
            ts.synchronize()
            dist.gather(d["hpack"], d["gbins"], dst=0)
        trace("step", i, "enqueued")

    def timed(n_warm, n_steps):
        """W untimed + exactly K timed steps, bracketed by barrier + synchronize; max over ranks."""
        for i in range(n_warm):
            step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i)
        torch.cuda.synchronize()
        el_own[0] = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            # every rank's own time for the K steps (taken before the closing barrier would hide it)
            t = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            t[rank] = el_own[0]
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            rank_seconds[:] = [float(v) for v in t.tolist()]
            t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    el_own, rank_seconds = [0.0], []
    dt = timed(args.warmup, args.steps)
    rank_seconds_headline = list(rank_seconds)
    trace("timed region done", dt)
    # what rank 0 gathered for its own shard must be what it decoded (the last step of every engine in flight)
    gather_checked = None
    if gather and rank == 0 and not os.environ.get("BENCH_DIAG_NO_DECODE"):
        n_kp, n_sc = B * MAX_PEOPLE * J * 3, B * MAX_PEOPLE
        for d in bufs[:min(S, args.steps + args.warmup)]:
            mine = d["gbins"][0].cpu()
            cnt = mine[n_kp + n_sc:].view(torch.int32)
            assert torch.equal(cnt, d["n"]), "gathered people counts differ from the rank-local decode"
            for b in range(B):
                m = min(int(cnt[b]), MAX_PEOPLE)
                assert torch.equal(mine[:n_kp].view(B, MAX_PEOPLE, J, 3)[b, :m], d["kp"][b, :m])
        gather_checked = "rank 0's own shard: gathered counts and keypoints equal the local decode"
    # short timed regions only (--steps < 100): a longer leg of the same steps (default 200, about 2 s) -- 0.2 s says
    # little about the rate the chip sustains once its clocks have settled.  Reported beside `value`.
    dt_sus = None
    if args.sustained_steps > 0 and args.steps < 100 and not os.environ.get("BENCH_DIAG_NO_DECODE"):
        dt_sus = timed(0, args.sustained_steps)
    # the PCIe-inclusive rate beside the headline (SURVEY 8d defines the metric with the H2D copy): same steps with
    # the frames copied from pinned host memory every step.  Reported, never `value`.
    dt_h2d = None
    if not args.h2d and not os.environ.get("BENCH_NO_H2D_LEG"):
        args.h2d = True
        dt_h2d = timed(min(args.warmup, 2), args.steps)
        args.h2d = False

    # Kernel durations for the roofline: the same steps once more on ONE engine, serialized, with a HIP event
    # around every op on its stream (under --streams > 1 the timed region overlaps kernels of different
    # batches, so per-kernel durations there would include the other batch's share of the chip).
    conv_ms_total = 0.0
    op_ms_total = np.zeros(len(plan.ops))
    decode_ms = np.zeros(3)
    n_prof = max(1, min(args.steps, 10)) if rank == 0 else 0
    pre_ms = 0.0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(n_prof):
        with torch.cuda.stream(tstreams[0]):      # events on the stream the kernel is launched on
            ev0.record()
        eng.preprocess(bufs[0]["frames"], _ffi.PH_PRE_NATIVE)
        with torch.cuda.stream(tstreams[0]):
            ev1.record()
        eng.forward_timed(B, wait=False)
        eng.decode_v2_dev(BODY_25, B, NET_H, NET_W, kp_scale, MAX_PEOPLE)
        tstreams[0].synchronize()
        ms = eng.forward_times()
        conv_ms_total += float(ms[conv_mask].sum())
        op_ms_total += ms
        decode_ms += np.array(eng.decode_v2_times())
        pre_ms += ev0.elapsed_time(ev1)
    # candidate pairs the PAF kernel scored (its algorithmic bytes depend on the peak counts)
    paf_pairs = 0
    if rank == 0:
        _, ex = eng.decode_v2(BODY_25, B, NET_H, NET_W, kp_scale, MAX_PEOPLE, want_peaks=True)
        npk = ex["peaks"][:, :, 0, 0].astype(np.int64)                      # [B, parts]
        paf_pairs = int(sum((npk[:, a] * npk[:, b]).sum() for a, b in BODY_25.pairs))

    if rank == 0 and args.per_op:
        ms = eng.forward_timed(B)
        fl = eng.op_flops() * B
        for name, o, t, f in zip(plan.op_names, plan.ops, ms, fl):
            print(f"{name:28s} {o['h']:4d}x{o['w']:<4d} k{o['k']} cin_read {o['cin_read']:4d} cout {o['cout']:4d} "
                  f"{t:8.4f} ms {f / max(t, 1e-9) / 1e9:9.1f} TFLOP/s", file=sys.stderr)
        print(f"total {ms.sum():.3f} ms, conv {ms[conv_mask].sum():.3f} ms", file=sys.stderr)

    if rank == 0:
        # opportunistic real-engine check (cv2.dnn on the COCO model against posehip, same images), never
        # a dependency: None when cv2, $POSEHIP_MODEL_DIR or $POSEHIP_SAMPLE_GLOB's images are absent
        from posehip import real_engine
        real = real_engine.run(os.environ.get("POSEHIP_MODEL_DIR", ""),
                               os.environ.get("POSEHIP_SAMPLE_GLOB", os.path.join(ROOT, "data", "sample", "*.jpg")))
        if real is None:
            print("real-engine parity: not available on this host (cv2 / Caffe model files absent)", file=sys.stderr)
        fps = world * B * args.steps / dt
        conv_s = conv_ms_total / 1e3
        achieved = flops_frame * B * n_prof / conv_s / 1e12 if conv_s > 0 else 0.0
        n_conv = int(conv_mask.sum())
        # dominant kernel = the instantiation with the largest share of the timed region; its
        # achieved rate = algorithmic FLOPs of its launches / their HIP-event durations
        kernels = eng.op_kernels()
        op_fl = eng.op_flops() * B
        by_k = {}
        for kname, t, f, c in zip(kernels, op_ms_total, op_fl, conv_mask):
            if c:
                a = by_k.setdefault(kname, [0.0, 0.0, 0])
                a[0] += t
                a[1] += f * n_prof
                a[2] += n_prof
        dom = max(by_k, key=lambda k: by_k[k][0])
        d_ms, d_fl, d_n = by_k[dom]
        d_ach = d_fl / (d_ms / 1e3) / 1e12 if d_ms > 0 else 0.0
        traffic = None
        tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_hbm_traffic.json") for r in (3, 2))
                      if os.path.exists(q)), "")
        if tpath:  # PMC passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE) of this same command
            with open(tpath) as fh:
                traffic = json.load(fh).get("bytes_per_launch", {}).get(dom)
        line = {
            "metric": "frames/sec (BODY_25 368x656, full PAF decode)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "BODY_25 batch-32 frames 368x656 per GPU: preprocess + backbone + "
                                   "PAF/heatmap stages + upsample/NMS + PAF matching + assembly "
                                   "(BASELINE configs[1])",
                       "batch_per_gpu": B, "net_resolution": f"{NET_W}x{NET_H}", "source_frames": f"{src_w}x{src_h}", "weights": "random He-normal",
                       "frames": "pinned host, H2D every step" if args.h2d else "resident in HBM",
                       "batches_in_flight_per_gpu": S, "hw_queues": hw_queues,
                       "parallelism": (f"frame-parallel x{world}, one RCCL gather of the packed keypoint record to "
                                       "rank 0 per step") if gather else "single GPU, no collective"},
            "roofline": {"bound": "mfma", "achieved": round(d_ach, 2), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(d_ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "kernel": dom,
                         "launches_per_step": d_n // n_prof,
                         "avg_launch_ms": round(d_ms / max(1, d_n), 4),
                         "algorithmic_gflop_per_launch": round(d_fl / max(1, d_n) / 1e9, 2),
                         "kernel_ms_per_step": round(d_ms / n_prof, 3),
                         "measured": f"HIP events per op, {n_prof} serialized steps on one engine after the timed region",
                         "all_conv": {"achieved": round(achieved, 2), "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                                      "launches_per_step": n_conv,
                                      "conv_ms_per_step": round(conv_ms_total / n_prof, 3),
                                      "algorithmic_gflop_per_frame": round(flops_frame / 1e9, 2),
                                      "per_kernel_ms_per_step": {k: round(v[0] / n_prof, 3)
                                                                 for k, v in sorted(by_k.items())}},
                         "traffic_source": f"profiles/{os.path.basename(tpath)}: FETCH_SIZE x2 + WRITE_SIZE of rocprofv3 PMC "
                                           "passes of this command on the profiling box (not measured in this run)",
                         "decode_ms_per_step": [round(float(x) / n_prof, 3) for x in decode_ms],
                         "tail": tail_block(B, src_h, src_w, pre_ms / n_prof, decode_ms / n_prof, paf_pairs),
                         # whole-pipeline view: frames/s x algorithmic FLOPs per frame against the MFMA peak of
                         # the GPUs used (100 % = 8 700 frames/s per GPU)
                         "end_to_end_frac_of_mfma_peak": round(fps * flops_frame / (world * MFMA_PEAK_TFLOPS * 1e12), 4)},
        }
        if dt_h2d is not None:
            line["h2d_inclusive"] = {"value": round(world * B * args.steps / dt_h2d, 2), "unit": "frames/s",
                                     "ms_per_step": round(dt_h2d / args.steps * 1e3, 3),
                                     "frac_of_resident": round(dt / dt_h2d, 4),
                                     "note": f"same {args.steps} steps with the u8 frames copied from pinned host "
                                             f"memory every step ({B * src_h * src_w * 3 / 1e6:.1f} MB/step over "
                                             "PCIe) by a copy stream one batch ahead of the engines; reported "
                                             "beside `value`, never as it"}
        if dt_sus is not None:
            line["sustained"] = {"value": round(world * B * args.sustained_steps / dt_sus, 2), "unit": "frames/s",
                                 "steps": args.sustained_steps,
                                 "ms_per_step": round(dt_sus / args.sustained_steps * 1e3, 3),
                                 "note": "the same step run for about 2 s right after the timed region"}
        if gather_checked:
            line["config"]["gather_check"] = gather_checked
        if real is not None:
            line["real_engine"] = real
        if world > 1 and rank_seconds_headline:
            # stragglers: every rank's own frames/s over the K steps (the headline divides by the slowest)
            per_rank = [B * args.steps / t for t in rank_seconds_headline]
            line["per_rank"] = {"unit": "frames/s", "min": round(min(per_rank), 2), "max": round(max(per_rank), 2),
                                "values": [round(v, 2) for v in per_rank]}
        if world == 1 and args.cpu_frames > 0:
            line["cpu_baseline"] = cpu_baseline(net, weights, BODY_25, args.cpu_frames, 1000)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
