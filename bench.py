#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + bounce rays) of the tracing hot path.

One step = one frame of the workload, traced by the HIP executor with the scene resident in
HBM.  With N > 1 (one process per GPU; `--gpus N` without a launcher starts the N ranks itself
through torch.distributed.run) the frame is split into interleaved 8-row tiles, every rank traces
its tiles, and the tiles are gathered to rank 0 over RCCL inside the timed region; total work is
fixed, so scaling is "strong".

`value` is the rate of a render loop that keeps K frames in flight (K executors of ONE device copy of the scene take the
frames in turn; every frame is traced completely and gathered); the line also carries `single_frame_ms` (one frame at a
time), `first_frame_ms` (a frame whose launches are sized from bounds only, as the first frame of a configuration is) and
`moving_camera` (the same loop with a new camera every frame).

Prints ONE JSON line on rank 0 with the whole-job rate and, at N = 1:
  roofline        the dominant kernel class of the default algorithm (hierarchy traversal): FLOP the kernel
                  executed (counted on the device) / its HIP-event time in frames rendered ONE AT A TIME on one stream
                  (launches that do not overlap) / vector-FP32 peak; the figures of the overlapped launches of the timed
                  region are kept beside it under `overlapped`;
  roofline_brute  the same frame traced by the reference's own algorithm (every ray against every triangle,
                  index order), 1 warm-up + 2 frames: executed packed-FP32 instructions / event time / peak;
  cpu_baseline    the CPU oracle (restatement of ray_tracer.cpp) on a bounded sample of the same frame, one
                  core and all cores.
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Hardware queues of this process (HIP's default is 4; read by the runtime when it starts): every executor with a frame in
# flight has a stream of its own (two with a side stream), torch and RCCL bring theirs.  Streams beyond the queues share one
# and their launches serialise; measured on one GPU, rank 0's share of C2 at G = 8: 0.50 ms with four frames in flight and 8
# queues, 0.48 ms with eight frames and 16 queues; with 4 queues and four frames 13 % slower than with 8 (whole frames: no
# difference).
# (an exported value is raised to what the run wants, never lowered: the GPU boxes export HIP's default of 4)
# (whole frames at N = 1 measure the same within 3 % for 4, 8 and 16 queues and 2 to 6 frames in flight, profiles/r03_sweep_queues_in_flight.txt:
# one GPU keeps the environment's value)
_world_env = int(os.environ.get("WORLD_SIZE", "1") or 1)
_want_queues = 16 if _world_env >= 4 else (8 if _world_env >= 2 else 0)
try:
    _have_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
except ValueError:
    _have_queues = 0
if os.environ.get("BENCH_HW_QUEUES"):                       # measurement aid: exactly this many
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["BENCH_HW_QUEUES"]
elif _have_queues < _want_queues:
    os.environ["GPU_MAX_HW_QUEUES"] = str(_want_queues)
os.environ.setdefault("HIPEXEC_KNOBS", "1")     # the library honours HIPEXEC_* knobs (HIPEXEC_NO_CULL of the brute-force leg) only with this set

F_TRI, F_SPH = 44.0, 25.0            # algorithmic FLOP per ray-triangle / ray-sphere test of the reference's scans
F_NODE = 121.0                       # FLOP per visited hierarchy node: growth factor 9 (bvh_node_growth) + four slab tests of 28 (bvh_box:
                                     # 3 centre - origin, 5 corner distance, 2 growth, 3 half extents, 3 products, 6 fused multiply-adds);
                                     # maxima, minima, comparisons and the stack carry none
PEAK_FP32_VALU_TFLOPS = 157.3        # MI355X vector FP32 peak (packed FMA: 256 CU x 4 SIMD x 16 lanes x 2 x 2 FLOP x 2.4 GHz)
# Packed-FP32 filter of the index-order scans, FLOP per lane: camera / lamp-line records take 3 packed multiplies
# (2 FLOP per lane each) + 6 packed FMAs (4 each) per PAIR of triangles and ray; the records for arbitrary rays
# 3 + 15.  A group that passes the filter is re-filtered per triangle (scalar form: 3 mul + 6 fma, or 3 + 15).
F_PAIR_CAM, F_PAIR_GEN = 3 * 2 + 6 * 4, 3 * 2 + 15 * 4
F_ONE_CAM, F_ONE_GEN = 3 * 1 + 6 * 2, 3 * 1 + 15 * 2

WORKLOADS = {
    # name: (scene builder, kwargs, width, height, max_depth, description)
    "c2": ("scene_c2", {}, 1920, 1080, 3, "C2: 100352-tri textured heightfield + 4 mirror spheres, 2 lamps, 1920x1080, 4 spp, depth-3 reflections"),
    "c3": ("scene_c2", {}, 3840, 2160, 5, "C3: C2 scene at 3840x2160, depth-5 reflections"),
    "c1": ("scene_c1", {}, 640, 480, 0, "C1: ~1k tris + 8 spheres, 1 lamp, 640x480, 4 spp, depth 0"),
    "c4": ("scene_c4", {}, 1920, 1080, 0, "C4: 10k tris, 8 x 2048^2 textures, 1920x1080, depth 0"),
    "c5": ("scene_c5", {}, 3840, 2160, 0, "C5: 20k tris + 16 spheres, 16 lamps, 3840x2160, depth 0"),
    "soup": ("scene_soup", {"seed": 105, "n_tri": 40000, "extent": 50.0}, 640, 480, 3,
             "soup: 40000 random triangles up to scene size (a few rays per frame outgrow the traversal stack), 640x480, depth 3"),
    "city": ("scene_city", {}, 1920, 1080, 2,
             "city (not a BASELINE config): 1500 boxes on a 96 x 96 ground grid, 36432 triangles, 2 lamps: walls, roofs and ground mix "
             "their orientations in every neighbourhood; 1920x1080, 4 spp, depth 2"),
    "city_yaw": ("scene_city", {"yaw": True}, 1920, 1080, 2,
                 "city_yaw (not a BASELINE config): the city's boxes turned by random angles (large triangles in every orientation), 1920x1080, depth 2"),
    "tiny": ("scene_c2", {"cells": 24, "tex_size": 64}, 256, 144, 3, "tiny: 1152-tri C2 variant at 256x144 (functional check only)"),
}

KNAMES = {"scan": ["k_primary", "k_closest", "k_anyhit_lamp", "shade/resolve"],
          "bvh": ["k_bvh_closest_pk", "k_bvh_closest", "k_bvh_shadow_direct", "shade/resolve"]}
KLABEL = ["camera rays", "bounce rays", "shadow rays", "shading"]


# ----------------------------------------------------------------------------------------
# launcher
# ----------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(args, argv):
    """`--gpus N` with N > 1 outside a launcher: start the N ranks (one process per GPU) and return their exit
    code; inside a launcher: check that its world size is the one asked for.  Runs before anything touches a GPU.
    Returns None when this process is a rank and should go on."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        if args.gpus <= 1:
            return None
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
        return subprocess.call(cmd)
    if int(world_env) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks", file=sys.stderr)
        return 2
    return None


# ----------------------------------------------------------------------------------------
# evidence from the committed profiles
# ----------------------------------------------------------------------------------------
def pmc_quote(kernel_class, workload, algo):
    """Counter-derived figures of a kernel class from the newest committed PMC summary of this algorithm
    (profiles/rNN_<algo>_pmc_summary.json, rocprofv3 --pmc passes on the C2 workload): VALU utilisation, live lanes
    per issued VALU instruction, L2 hit rate, resident waves per SIMD, HBM bytes per launch (FETCH_SIZE with the
    gfx950 x2 correction + WRITE_SIZE).  None when no summary applies."""
    if workload != "c2":
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{algo}_pmc_summary.json")))
    if not files:
        return None
    try:
        summ = json.load(open(files[-1]))
    except Exception:
        return None
    for name, m in summ.items():
        if not name.startswith("rt::" + kernel_class + "<"):
            continue
        c = m.get("counters_per_frame") or m.get("counters", {})      # ratios below: any common normalisation cancels
        q = {"source": os.path.relpath(files[-1], ROOT), "kernel": name}
        for k in ("l2_hit_rate", "occupancy_waves_per_simd"):
            if k in m:
                q[k] = m[k]
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU"):
            # fraction of the VALU issue peak: a SIMD (32 lanes wide) issues one wave64 instruction per 2 cycles;
            # GRBM_GUI_ACTIVE is the sum over the 8 XCDs, 1024 SIMDs on the chip
            q["valu_issue_fraction"] = c["SQ_INSTS_VALU"] * 2.0 / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
            # rocprofiler-sdk's derived metrics for gfx950 (counter_defs.yaml): VALUBusy = 100 * SQ_ACTIVE_INST_VALU / CU_NUM /
            # GRBM_GUI_ACTIVE (per XCD), VALUUtilization = 100 * SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
            q["rocprof_VALUBusy_pct"] = 100.0 * c["SQ_ACTIVE_INST_VALU"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
            q["rocprof_VALUUtilization_pct"] = 100.0 * c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        if c.get("SQ_ACTIVE_INST_VALU"):
            q["live_lane_fraction"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        if c.get("SQ_WAVE_CYCLES"):
            q["wait_fraction_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if "hbm_bytes_per_launch" in m:
            q["hbm_bytes_per_launch"] = m["hbm_bytes_per_launch"]
        for k in ("l2_miss_bytes_per_launch", "avg_launch_ms_serial", "valu_insts_per_wave"):
            if k in m:
                q[k] = m[k]
        return q
    return None


def limiter_of(q, packed_frac=0.0):
    """What the counters say holds the kernel back (the roofline's unit stays FLOP/s against the VALU peak).
    `packed_frac`: share of the FP32 peak reached through packed instructions alone (the brute-force filter): a packed
    instruction holds the SIMD for two issue slots and the counter counts it once, so half of that share is added."""
    if not q or "valu_issue_fraction" not in q:
        return None
    if packed_frac:
        q["valu_issue_fraction_packed_counted_twice"] = q["valu_issue_fraction"] + packed_frac / 2.0
        q = dict(q, valu_issue_fraction=q["valu_issue_fraction_packed_counted_twice"])
    if q["valu_issue_fraction"] >= 0.8:
        return "VALU issue (the SIMDs issue vector instructions at >= 80 % of their peak rate)"
    return ("latency and divergence: the SIMDs issue VALU instructions at %.0f %% of their peak rate (one wave64 instruction "
            "per 2 cycles; rocprof's VALUBusy, which prices an instruction at 4 cycles: %.0f %%) with %.0f %% of the lanes live, "
            "%.1f waves per SIMD resident; neither HBM nor FP32 throughput is saturated"
            % (100 * q["valu_issue_fraction"], q.get("rocprof_VALUBusy_pct", 0.0), 100 * q.get("live_lane_fraction", 0.0),
               q.get("occupancy_waves_per_simd", 0.0)))


# ----------------------------------------------------------------------------------------
# FLOP models
# ----------------------------------------------------------------------------------------
def scan_equiv_flops(st, k, n_sph):
    """Algorithmic FLOP of a ray class under the reference's linear scans: every triangle and sphere per
    closest-hit ray; up to the first occluder per shadow ray."""
    if k == 0:
        return st.tests_tri_primary * F_TRI + st.rays_primary * n_sph * F_SPH
    if k == 1:
        return (st.tests_tri_closest - st.tests_tri_primary) * F_TRI + st.rays_bounce * n_sph * F_SPH
    if k == 2:
        return st.tests_tri_any * F_TRI + st.tests_sph_any * F_SPH
    return 0.0


def executed_flops_bvh(st, k, n_sph):
    """FLOP the hierarchy kernels evaluated for a ray class (node visits + exact tests + sphere tests, counted on the device)."""
    if k >= 3:
        return 0.0
    rays = [st.rays_primary, st.rays_bounce, st.rays_shadow][k]
    return st.bvh_nodes[k] * F_NODE + st.bvh_tests[k] * F_TRI + rays * n_sph * F_SPH


def executed_flops_scan(st, k):
    """FLOP of the packed-FP32 filter instructions the scan kernels executed for a ray class: per (wave, scanned
    128-triangle chunk) 64 pairs x 64 lanes x rays per lane x the pair filter; per (wave, group that passed) 4 triangles
    re-filtered one by one.  The exact tests of the few survivors (44 FLOP each) are left out, so this undercounts."""
    if k >= 3:
        return 0.0
    rays_per_lane = [4, int(os.environ.get("HIPEXEC_CLOSEST_R", "3")), int(os.environ.get("HIPEXEC_ANYHIT_R", "2"))][k]
    f_pair, f_one = [(F_PAIR_CAM, F_ONE_CAM), (F_PAIR_GEN, F_ONE_GEN), (F_PAIR_CAM, F_ONE_CAM)][k]
    per_lane = st.scanned_chunks[k] * 64.0 * f_pair + st.exact_groups[k] * 4.0 * f_one
    return per_lane * 64.0 * rays_per_lane


# ----------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(fs, W, H, depth, budget_s=20.0):
    """CPU oracle (oracle/rt_oracle.c, a restatement of ray_tracer.cpp built with the reference's flags) on a regular
    grid of 8x8-pixel blocks of the frame: one thread, then a block-parallel pool like renderParallel
    (ray_tracer.cpp:57-153) on up to 16 threads."""
    import numpy as np
    from oracle.binding import Oracle
    threads = max(1, min(16, os.cpu_count() or 1))
    orc = Oracle(fs)

    def blocks(n):
        g = max(1, int(round(n ** 0.5)))
        xs = np.linspace(0, W - 8, g).astype(int) // 8 * 8
        ys = np.linspace(0, H - 8, g).astype(int) // 8 * 8
        return np.array([(x, y, 8, 8) for y in ys for x in xs], dtype=np.int32)

    def timed(n_blocks, n_threads):
        b = blocks(n_blocks)
        t0 = time.perf_counter(); st = orc.render_blocks(W, H, b, max_depth=depth, n_threads=n_threads); dt = time.perf_counter() - t0
        return st.rays, dt, b.shape[0]

    def leg(n_threads, budget):
        rays, dt, nb = timed(max(4, n_threads), n_threads)                       # probe sizes the sample
        n = int(min(4096, max(n_threads, budget * (rays / max(dt, 1e-9)) / max(1.0, rays / nb))))
        rays, dt, nb = timed(n, n_threads)
        return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": n_threads,
                "sample": f"{nb} blocks of 8x8 pixels on a regular grid of the {W}x{H} frame, full depth ({rays} rays, {dt:.1f} s)"}

    one = leg(1, 0.3 * budget_s)
    allc = leg(threads, 0.7 * budget_s)
    allc.update({"kind": "port",
                 "kind_note": "oracle restatement of ray_tracer.cpp (oracle/rt_oracle.c, -O2, no -march, no fast-math, block-parallel "
                              "pthread pool); the reference itself cannot be compiled here (its glm dependency is absent)",
                 "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "one_core": one})
    return allc


# ----------------------------------------------------------------------------------------
# per-rank diagnostics of an N > 1 run
# ----------------------------------------------------------------------------------------
class RankDiag:
    """What one rank spent per frame of the timed region, so that a scaling shortfall can be read off the line: the device
    time of its share of the frame (the executor's HIP events around the frame's launches), the time this process
    waited for it in collect, and the gather (events on the stream the collective runs on; on rank 0 that includes the
    de-interleave copy).  gather() brings every rank's means to rank 0."""
    FIELDS = ("compute_ms", "host_wait_ms", "gather_ms", "tiles")

    def __init__(self, frames_in_flight, hw_queues):
        self.sum = [0.0] * len(self.FIELDS)
        self.frames = 0
        self.frames_in_flight, self.hw_queues = frames_in_flight, hw_queues

    def add(self, compute_ms, host_wait_ms, gather_ms, tiles):
        for k, v in enumerate((compute_ms, host_wait_ms, gather_ms, tiles)):
            self.sum[k] += float(v)
        self.frames += 1

    def gather(self, rank, world, device):
        import torch
        import torch.distributed as dist
        mine = torch.tensor([float(rank), float(self.frames), float(self.frames_in_flight), float(self.hw_queues)]
                            + [v / max(1, self.frames) for v in self.sum], dtype=torch.float64, device=device)
        rows = [mine]
        if world > 1:
            rows = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
        if rank != 0:
            return None
        out = []
        for r in rows:
            v = [float(x) for x in r.cpu()]
            d = {"rank": int(v[0]), "frames": int(v[1]), "frames_in_flight": int(v[2]), "hw_queues": int(v[3])}
            d.update({name: v[4 + k] for k, name in enumerate(self.FIELDS)})
            out.append(d)
        return out


# ----------------------------------------------------------------------------------------
# functional self-test of the multi-process path on CPU (gloo): no GPU, no executor, no rate
# ----------------------------------------------------------------------------------------
def selftest_cpu(args, rank, world):
    """The launcher, rendezvous, tile partition, gather and max-over-ranks timing of this file with a stand-in for
    the executor that writes each tile's frame index into its rows.  Prints a line marked "selftest": true."""
    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import gather_buffer, gather_frame, local_tiles, max_local_tiles
    _, _, W, H, _, desc = WORKLOADS[args.workload]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tmax = max_local_tiles(H, world)
    out = torch.zeros((tmax * 8, W, 4), dtype=torch.float32)
    gbuf = gather_buffer(out)
    ok = True
    diag = RankDiag(1, 1)
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        tc = time.perf_counter()
        for j in range(local_tiles(H, rank, world)):
            out[8 * j:8 * j + 8] = float(rank + j * world)
        tg = time.perf_counter()
        frame = gather_frame(out, H, buf=gbuf)
        diag.add((tg - tc) * 1e3, (tg - tc) * 1e3, (time.perf_counter() - tg) * 1e3, local_tiles(H, rank, world))
        if rank == 0:
            ok = ok and bool(torch.equal(frame[:, 0, 0], (torch.arange(H) // 8).float()))
    elapsed = time.perf_counter() - t0
    seen = torch.ones(1)
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(seen); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_rank = diag.gather(rank, world, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"selftest": True, "ok": ok, "n_gpus": world, "ranks_seen": int(seen.item()), "steps": args.steps,
                          "warmup": args.warmup, "elapsed_s": float(t[0]), "data": "stand-in executor on CPU (gloo): not a measurement",
                          "config": {"workload": desc}, "per_rank": per_rank}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return 0 if ok else 1


# ----------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--algo", default="bvh", choices=["bvh", "scan", "brute"],
                    help="bvh: hierarchy traversal (default); scan: index-order scans with chunk culling; "
                         "brute: index-order scans of every triangle, the reference's own work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-brute", action="store_true", help="leave out the roofline_brute leg (1 warm-up + 2 frames of full scans)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--lanes-per-frame", type=int, default=0, help="lanes of each executor (0: 1 with frames in flight, else 2)")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=list(range(0, 9)),
                    help="executors that take the frames in turn (hierarchy algorithm; 1: one executor with two lanes per frame)")
    ap.add_argument("--fast-build", action="store_true",
                    help="keep the hierarchy hipexec_create builds on the device (Morton order, under a millisecond) instead of "
                         "rebuilding it once with the surface-area topology (hipexec_optimize)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="leave out the serial / single-frame / first-frame / moving-camera legs (profiling passes: every frame of the "
                         "run is then a frame of the timed loop's kind)")
    ap.add_argument("--selftest-cpu", action="store_true", help="functional check of the multi-process path on CPU (gloo), no GPU")
    args = ap.parse_args(argv)

    rc = launch_ranks_if_needed(args, argv)
    if rc is not None:
        return rc
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)

    import torch
    import torch.distributed as dist
    from raytracer_amd import _abi, scenegen
    from raytracer_amd.dist import gather_buffer, gather_frame, max_local_tiles
    from raytracer_amd.executor import FramesInFlight, HipExecutor

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the HIP path)")
    # BENCH_ONE_GPU_REHEARSAL=1: every rank on GPU 0 and gloo instead of RCCL (whose ranks need a GPU each), to run the N > 1
    # code path of this file on a box with one GPU.  Its line is marked as a rehearsal and is not a measurement.
    rehearsal = world > 1 and os.environ.get("BENCH_ONE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if rehearsal else dev          # where the small reductions live (gloo: host tensors)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    builder, kw, W, H, depth, desc = WORKLOADS[args.workload]
    fs = getattr(scenegen, builder)(**kw).to_flat()
    if args.algo == "brute":
        os.environ["HIPEXEC_NO_CULL"] = "1"
    algo = "bvh" if args.algo == "bvh" else "scan"
    # Throughput mode: several executors of the scene, one lane each, take the frames in turn, so that the thinly populated
    # deep recursion levels of one frame run beside the large launches of the next (the scans gain nothing from it).
    # (three at N = 1: C2 3.14 ms with two, 3.09 with three, 3.10 with four; four for the shares of N = 2 and 3, eight for the
    # small shares from N = 4 up: rank 0's share of C2 at G = 4 0.91 -> 0.88 ms, at G = 8 0.50 -> 0.48 ms)
    in_flight = (args.frames_in_flight or (3 if world == 1 else 4 if world < 4 else 8)) if algo == "bvh" else 1
    # The scene is kept for the whole run, so its hierarchy is rebuilt once with the surface-area topology (same frames bit
    # for bit, fewer node visits; outside the timed region like the rest of the scene set-up, its cost is in the line).
    pipe = FramesInFlight(fs, device=local_rank, depth=in_flight, lanes=args.lanes_per_frame or None,
                          optimize=(algo == "bvh" and not args.fast_build))
    os.environ.pop("HIPEXEC_NO_CULL", None)
    ex = pipe.ex[0]
    tmax = max_local_tiles(H, world)
    outs = [torch.zeros((tmax * 8, W, 4), dtype=torch.float32, device=dev) for _ in range(in_flight)]
    gbufs = [gather_buffer(outs[k]) for k in range(in_flight)]       # rank 0's receive buffers, one per slot, made once
    st = _abi.Stats()
    frame = None
    frames_rendered = {"bvh": 0, "scan": 0}
    diag = None                               # RankDiag while the timed region runs

    gathered = [None] * in_flight             # per slot: event on torch's stream after the gather that last read outs[slot]
    g_begin = [torch.cuda.Event(enable_timing=True) for _ in range(in_flight)]
    g_end = [torch.cuda.Event(enable_timing=True) for _ in range(in_flight)]
    g_pending = []                            # slots whose gather times have not been read yet

    def submit(time_kernels, a=None, camera=None):
        k = pipe.next
        if gathered[k] is not None:           # the executor writes outs[k] from its own streams: not before that gather has read it
            gathered[k].synchronize()
            gathered[k] = None
            drain_gather_times(k)             # the slot's events are about to be recorded again
        if camera is not None:
            pipe.ex[k].set_camera(*camera)
        frames_rendered[a or algo] += 1
        return pipe.submit(W, H, outs[k].data_ptr(), None, max_depth=depth, tone=True, tile_first=rank, tile_step=world,
                           time_kernels=time_kernels, algo=a or algo)

    def drain_gather_times(only=None):
        # gather durations of collected frames whose events have completed (torch's stream was synchronised past them)
        for e in [e for e in g_pending if only is None or e[0] == only]:
            g_pending.remove(e)
            if diag is not None:
                diag.add(e[1], e[2], g_begin[e[0]].elapsed_time(g_end[e[0]]), tmax)

    def collect(slot):
        nonlocal frame
        t_wait = time.perf_counter()
        s = pipe.collect(slot)                       # waits for the frame's kernels (the executor's own streams)
        t_wait = (time.perf_counter() - t_wait) * 1e3
        g_begin[slot].record()
        frame = gather_frame(outs[slot], H, buf=gbufs[slot])   # RCCL gather to rank 0 + de-interleave (torch's stream)
        g_end[slot].record()
        gathered[slot] = g_end[slot]
        if diag is not None:
            g_pending.append((slot, float(s.total_ms), t_wait))
        return s

    def step(time_kernels, exe=None, a=None):
        """One synchronous frame: on the pipeline's next executor, or on `exe` (the brute-force leg's executor)."""
        nonlocal frame
        if exe is None:
            s = collect(submit(time_kernels, a))
            C.memmove(C.byref(st), C.byref(s), C.sizeof(_abi.Stats))
            return
        frames_rendered[a or algo] += 1
        exe.render_device(W, H, outs[0].data_ptr(), None, max_depth=depth, tone=True, tile_first=rank, tile_step=world,
                          time_kernels=time_kernels, stats=st, algo=a or algo)
        exe.sync()
        frame = gather_frame(outs[0], H, buf=gbufs[0])

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The test counts of the reference's scans are a property of the frame; only the index-order scans can count the
    # shadow-ray part, so they come from one untimed scan frame (reported as scan_equivalent, never as the roofline).
    ref_flops = None
    if algo == "bvh" and rank == 0 and world == 1:
        step(False, a="scan")
        ref_flops = [scan_equiv_flops(st, k, fs.n_sph) for k in range(3)]
    # One untimed frame with every kernel class timed picks the dominant class; the timed steps then bracket only that
    # class's launches with events (every timed launch costs two event records on its stream).
    # every executor of the pipeline renders once before anything is timed (its buffers and its remembered queue sizes
    # belong to it): the W warm-up steps, and more where W is smaller than the number of executors
    # (twice through the executors: an executor sizes its launches exactly once two frames of the configuration agreed)
    for _ in range(max(args.warmup, 2 * in_flight)):
        step(False)
    step(True)
    probe_ms = [float(st.kernel_ms[k]) for k in range(4)]
    dom = max(range(3), key=lambda k: probe_ms[k])
    agg = {k: 0 for k in ("rays", "ms", "launches", "exec_flop", "equiv_flop", "exec_flop_all", "retraced")}
    walk = {"rays": [0, 0, 0], "nodes": [0, 0, 0], "tests": [0, 0, 0], "overflows": 0}      # per ray class: camera, bounce, shadow

    def account(s):
        agg["rays"] += s.rays
        agg["retraced"] += int(s.retraced)
        agg["ms"] += s.kernel_ms[dom]; agg["launches"] += s.launches[dom]
        agg["exec_flop"] += executed_flops_bvh(s, dom, fs.n_sph) if algo == "bvh" else executed_flops_scan(s, dom)
        agg["exec_flop_all"] += sum((executed_flops_bvh(s, k, fs.n_sph) if algo == "bvh" else executed_flops_scan(s, k)) for k in range(3))
        agg["equiv_flop"] += ref_flops[dom] if ref_flops is not None else scan_equiv_flops(s, dom, fs.n_sph)
        for k, r in enumerate((s.rays_primary, s.rays_bounce, s.rays_shadow)):
            walk["rays"][k] += int(r); walk["nodes"][k] += int(s.bvh_nodes[k]); walk["tests"][k] += int(s.bvh_tests[k])
        walk["overflows"] += int(s.bvh_stack_overflows)


    def timed_loop(n_steps, time_classes, cameras=None, on_stats=None):
        """n_steps frames submitted in_flight deep, bracketed by fences; -> elapsed seconds."""
        fence()
        t_start = time.perf_counter()
        slots = []
        for i in range(n_steps):
            if len(slots) >= in_flight:
                sk = collect(slots.pop(0))
                if on_stats:
                    on_stats(sk)
            slots.append(submit(time_classes, camera=cameras[i] if cameras else None))
        while slots:
            sk = collect(slots.pop(0))
            if on_stats:
                on_stats(sk)
        fence()
        return time.perf_counter() - t_start

    diag = RankDiag(in_flight, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    elapsed = timed_loop(args.steps, [dom], on_stats=account)
    drain_gather_times()
    rank_diag, diag = diag, None
    lanes = int(st.lanes)
    per_rank = rank_diag.gather(rank, world, red_dev) if world > 1 else None

    # ------------------------------------------------------------------------------------------------------------
    # further legs at N = 1 (hierarchy algorithm): what the headline number is not
    # ------------------------------------------------------------------------------------------------------------
    extra = {}
    if world == 1 and algo == "bvh" and not args.no_extra_legs:
        n_leg = max(3, min(10, args.steps))
        # (a) frames ONE AT A TIME on one stream of one executor, every launch timed: launches that do not overlap.
        #     This is what the roofline fraction of a kernel class is priced on.
        ser = {"ms": [0.0] * 4, "launches": [0] * 4, "flop": [0.0] * 3, "frame_ms": 0.0}
        e0 = pipe.ex[0]
        fence()
        for _ in range(n_leg):
            frames_rendered["bvh"] += 1
            e0.render_device(W, H, outs[0].data_ptr(), None, max_depth=depth, tone=True, time_kernels=True, stats=st, algo="bvh")
            e0.sync()
            ser["frame_ms"] += float(st.total_ms)
            for k in range(4):
                ser["ms"][k] += float(st.kernel_ms[k]); ser["launches"][k] += int(st.launches[k])
            for k in range(3):
                ser["flop"][k] += executed_flops_bvh(st, k, fs.n_sph)
        ser["frames"] = n_leg
        extra["serial"] = ser

        def wall_per_frame(exe, n):
            torch.cuda.synchronize(dev)
            t_a = time.perf_counter()
            for _ in range(n):
                frames_rendered["bvh"] += 1
                exe.render_device(W, H, outs[0].data_ptr(), None, max_depth=depth, tone=True, stats=st, algo="bvh")
                exe.sync()
            return (time.perf_counter() - t_a) / n * 1e3

        # (b) one frame at a time the way a single caller gets it: one executor, two lanes (the library's default)
        two = pipe.ex[0].clone()
        try:
            two.set_lanes(2)
            wall_per_frame(two, 3)
            extra["single_frame_ms"] = wall_per_frame(two, n_leg)
        finally:
            two.close()
        # (c) a frame none of whose launch sizes is remembered: every level launched for the bound n(L+1) <= n(L), the way
        #     the first frame of a configuration (or of a new scene) is; the executor's buffers exist (one frame before)
        os.environ["HIPEXEC_PREDICT"] = "2"
        try:
            fresh = pipe.ex[0].clone()
        finally:
            os.environ.pop("HIPEXEC_PREDICT", None)
        try:
            fresh.set_lanes(2)
            wall_per_frame(fresh, 1)
            extra["first_frame_ms"] = wall_per_frame(fresh, n_leg)
        finally:
            fresh.close()
        # (d) the timed loop again with a new camera every frame (a slow orbit and dolly): nothing is carried from frame to
        #     frame except launch-size estimates, so the rate should be that of the static loop
        import numpy as np
        cams = []
        for i in range(args.steps):
            a = 0.004 * (i + 1)
            ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
            cams.append((fs.cam_pos + np.float32(0.01 * (i + 1)) * np.array([1.0, 0.1, -0.5], np.float32), (fs.cam_mat @ ry).astype(np.float32)))
        mv = {"rays": 0, "retraced": 0}

        def mv_account(sk):
            mv["rays"] += sk.rays
            mv["retraced"] += int(sk.retraced)
        timed_loop(min(args.steps, 2 * in_flight), False, cameras=cams)            # estimates adapt to a moving camera first
        mv_elapsed = timed_loop(args.steps, False, cameras=cams, on_stats=mv_account)
        extra["moving_camera"] = {"ms_per_step": mv_elapsed / args.steps * 1e3, "value": mv["rays"] / mv_elapsed / 1e6, "unit": "Mrays/s",
                                  "frames_retraced": mv["retraced"],
                                  "note": "the same loop with a new camera before every frame (hipexec_set_camera: 0.23 degrees of orbit and a "
                                          "small dolly per frame); every frame is traced completely"}
        for e in pipe.ex:
            e.set_camera(fs.cam_pos, fs.cam_mat)

    t = torch.tensor([elapsed, float(agg["rays"]), 1.0], dtype=torch.float64, device=red_dev)
    if world > 1:
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed_max, rays_total, ranks_seen = float(tm[0]), float(ts[1]), int(round(float(ts[2])))
    else:
        elapsed_max, rays_total, ranks_seen = elapsed, float(agg["rays"]), 1
    if ranks_seen != world:
        raise SystemExit(f"all-reduce saw {ranks_seen} ranks, expected {world}")

    if rank == 0:
        names = KNAMES[algo]
        tf = lambda f, ms: f / max(ms, 1e-9) / 1e9                       # FLOP, ms -> TFLOP/s
        ach_overlapped = tf(agg["exec_flop"], agg["ms"])
        ser = extra.get("serial")
        if ser:      # launches that do not overlap: frames one at a time on one stream
            ach = tf(ser["flop"][dom], ser["ms"][dom])
            avg_launch_ms, launches_per_step = ser["ms"][dom] / max(1, ser["launches"][dom]), ser["launches"][dom] / ser["frames"]
            basis = (f"{ser['frames']} frames rendered one at a time on one stream after the timed region (every launch timed with HIP events): "
                     "launches that do not overlap")
        else:        # no serial leg (N > 1, scan algorithms, --no-extra-legs): the timed region's own launches
            ach, avg_launch_ms, launches_per_step = ach_overlapped, agg["ms"] / max(1, agg["launches"]), agg["launches"] / args.steps
            basis = "launches of the timed region" + (" (frames in flight: they overlap)" if in_flight > 1 else "")
        q = pmc_quote(names[dom], args.workload, args.algo)
        roof = {"bound": "valu", "kernel": f"{names[dom]} ({KLABEL[dom]})", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / PEAK_FP32_VALU_TFLOPS, "basis": basis,
                "traffic": q.get("hbm_bytes_per_launch") if q else None,
                "flop_model": (f"work the kernel executed, counted on the device: {F_NODE:.0f} FLOP per visited 4-wide node + 44 per exact "
                               "triangle test + 25 per sphere test") if algo == "bvh" else
                              ("packed-FP32 filter instructions the kernel executed (30 or 66 FLOP per lane, triangle pair and ray; "
                               "packed FMA = 4 FLOP per lane), exact tests of survivors left out"),
                "flop_model_note": ("a fraction of the FLOP peak rewards instructions, not results: shortening the node step from 190 to 121 FLOP "
                                    "made the frame faster and this fraction smaller; the kernels are bound by their vector-instruction count "
                                    "(see pmc: rocprof VALUBusy)") if algo == "bvh" else None,
                "avg_launch_ms": avg_launch_ms, "launches_per_step": launches_per_step,
                "kernel_ms_per_step": ({f"{names[k]} ({KLABEL[k]})": ser["ms"][k] / ser["frames"] for k in range(4)} if ser else
                                       {f"{names[k]} ({KLABEL[k]})": probe_ms[k] for k in range(4)}),
                "kernel_ms_note": ("every class, launches that do not overlap (the serial frames); their sum is the device time of one frame alone on "
                                   "one stream" if ser else "all classes: one untimed frame with every launch timed"),
                "serial_frame_ms": ser["frame_ms"] / ser["frames"] if ser else None,
                "class_frac": ({f"{names[k]} ({KLABEL[k]})": tf(ser["flop"][k], ser["ms"][k]) / PEAK_FP32_VALU_TFLOPS for k in range(3)} if ser else None),
                "lanes": lanes, "frames_in_flight": in_flight,
                "overlapped": {"avg_launch_ms": agg["ms"] / max(1, agg["launches"]), "launches_per_step": agg["launches"] / args.steps,
                               "achieved": ach_overlapped, "frac": ach_overlapped / PEAK_FP32_VALU_TFLOPS,
                               "note": "the same class's launches inside the timed region, where launches of the frames in flight share the chip: each "
                                       "takes longer than alone and their durations sum to more than the wall time; not a serial quantity"},
                "whole_frame": {"achieved": agg["exec_flop_all"] / max(elapsed, 1e-12) / 1e12, "frac": agg["exec_flop_all"] / max(elapsed, 1e-12) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                                "note": "executed FLOP of all three ray classes of this rank / wall time of the timed region: what the chip "
                                        "delivered while launches of several frames shared it (the per-launch figure above divides by "
                                        "durations that overlap)"},
                "unfused_ceiling": {"tflops": PEAK_FP32_VALU_TFLOPS / 2.0,
                                    "note": "the peak counts a fused multiply-add as 2 FLOP per lane and issue slot; the exact arithmetic of the "
                                            "path must round after every multiply and every add like the reference's x86 build "
                                            "(-ffp-contract=off), 1 FLOP per lane and slot, so half the peak is the most this code could reach "
                                            "with every lane live and every slot carrying a FLOP; compares, min/max and selects (about 40 % of "
                                            "a node step) carry none"},
                "walk_per_ray": ({KLABEL[k]: {"nodes": walk["nodes"][k] / max(1, walk["rays"][k]), "exact_tests": walk["tests"][k] / max(1, walk["rays"][k])}
                                  for k in range(3)} if algo == "bvh" else None),
                "stack_overflows_per_step": walk["overflows"] / args.steps,
                "limiter": limiter_of(q, ach / PEAK_FP32_VALU_TFLOPS if algo != "bvh" else 0.0), "pmc": q,
                "scan_equivalent": {"tflops": tf(agg["equiv_flop"], agg["ms"]),
                                    "note": "FLOP the reference's linear scans would have spent on the same rays (44 per ray-triangle test, every "
                                            "triangle per closest-hit ray, up to the first occluder per shadow ray) / the same kernel time: an "
                                            "algorithm ratio, not a hardware fraction"}}
        if lanes > 1 or in_flight > 1:
            roof["lanes_note"] = ("launches of concurrent streams overlap (frames in flight on an executor each, or the two lanes of one "
                                  "frame); kernel times are per-launch HIP-event durations and sum to more than ms_per_step")
        res = {
            "metric": "Mrays/sec (primary+secondary)", "value": rays_total / elapsed_max / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL: all ranks on one GPU, gloo through host memory: not a measurement",
            "config": {"workload": desc, "width": W, "height": H, "max_depth": depth, "triangles": fs.n_tri,
                       "spheres": fs.n_sph, "lamps": fs.n_lamp, "mode": "cpu-compat (ray_tracer.cpp semantics)",
                       "algorithm": args.algo, "frames_in_flight": in_flight,
                       "hierarchy": ("index-order scans, no hierarchy" if algo != "bvh" else
                                     "Morton-order build of hipexec_create (device, < 1 ms)" if args.fast_build else
                                     f"surface-area topology, hipexec_optimize once per scene: {pipe.optimize_ms:.1f} ms (host cut + device refit), "
                                     "outside the timed region like the scene upload; --fast-build keeps hipexec_create's Morton-order tree"),
                       "what_a_step_is": (f"one complete frame; ms_per_step is the period of a loop that keeps {in_flight} frames in flight on "
                                          f"{in_flight} executors of one device copy of the scene (the reference's timing loop, main.cpp:92-105, "
                                          "also repeats one static frame); single_frame_ms is one frame at a time, first_frame_ms a frame with "
                                          "no remembered launch sizes, moving_camera the loop with a new camera every frame"),
                       "partition": f"interleaved 8-row tiles over {world} GPU(s), gather to rank 0",
                       "rays_per_frame": rays_total / args.steps},
            "frames_rendered": frames_rendered,
            "frames_retraced": agg["retraced"],      # frames of the timed region traced twice because a launch was sized too small
            "roofline": roof,
        }
        for k in ("single_frame_ms", "first_frame_ms", "moving_camera"):
            if k in extra:
                res[k] = extra[k]
        if per_rank is not None:
            res["per_rank"] = per_rank
        if world == 1 and algo == "bvh" and not args.no_brute:
            res["roofline_brute"] = brute_leg(fs, local_rank, step, st, args.workload)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(fs, W, H, depth, args.cpu_budget)
        print(json.dumps(res), flush=True)
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def brute_leg(fs, device, step, st, workload, warmup=1, steps=2):
    """The same frame by the reference's algorithm: index-order scans of every triangle (no chunk is skipped), every
    launch timed with HIP events.  Returns the roofline of its dominant kernel class from executed instructions."""
    from raytracer_amd.executor import HipExecutor
    os.environ["HIPEXEC_NO_CULL"] = "1"
    try:
        exb = HipExecutor(fs, device=device)
    finally:
        os.environ.pop("HIPEXEC_NO_CULL", None)
    try:
        for _ in range(warmup):
            step(False, exe=exb, a="scan")
        ms = [0.0] * 4; launches = [0] * 4; flop = [0.0] * 3; equiv = [0.0] * 3; rays = 0; total_ms = 0.0
        for _ in range(steps):
            step(True, exe=exb, a="scan")
            rays += st.rays; total_ms += st.total_ms
            for k in range(4):
                ms[k] += st.kernel_ms[k]; launches[k] += st.launches[k]
            for k in range(3):
                flop[k] += executed_flops_scan(st, k); equiv[k] += scan_equiv_flops(st, k, fs.n_sph)
    finally:
        exb.close()
    names = KNAMES["scan"]
    dom = max(range(3), key=lambda k: ms[k])
    tf = lambda f, m: f / max(m, 1e-9) / 1e9
    q = pmc_quote(names[dom], workload, "brute")
    ach = tf(flop[dom], ms[dom])
    return {"bound": "valu", "kernel": f"{names[dom]} ({KLABEL[dom]})", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_FP32_VALU_TFLOPS, "traffic": q.get("hbm_bytes_per_launch") if q else None,
            "flop_model": "packed-FP32 filter instructions the kernel executed, from its own counters: per (wave, 128-triangle chunk) 64 pairs x 64 lanes "
                          "x rays per lane x (3 packed mul + 6 or 15 packed FMA; packed FMA = 4 FLOP per lane); min/max/compare instructions "
                          "and the exact tests of survivors carry no FLOP, so the fraction stays below the VALU issue fraction",
            "avg_launch_ms": ms[dom] / max(1, launches[dom]), "steps": steps, "warmup": warmup,
            "ms_per_frame": total_ms / steps, "mrays_per_s": rays / max(total_ms, 1e-9) / 1e3,
            "kernel_ms_per_frame": {f"{names[k]} ({KLABEL[k]})": ms[k] / steps for k in range(4)},
            "kernel_frac": {f"{names[k]} ({KLABEL[k]})": tf(flop[k], ms[k]) / PEAK_FP32_VALU_TFLOPS for k in range(3)},
            "limiter": limiter_of(q, ach / PEAK_FP32_VALU_TFLOPS), "pmc": q,
            "scan_equivalent": {"tflops": tf(equiv[dom], ms[dom]),
                                "note": "44 FLOP per ray-triangle test of the reference's scalar code; the filter answers most tests with "
                                        "fewer instructions, so this exceeds what the silicon executed"}}


if __name__ == "__main__":
    sys.exit(main())
