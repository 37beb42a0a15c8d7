"""Two processes on ONE GPU, each rendering one half of the frames' tiles (tile_first = its index, tile_step = 2) with K
frames in flight and L lanes per frame: what queue contention between processes looks like where only one GPU is at hand.
python tools/two_ranks_one_gpu.py c2     (parent: starts the workers for several settings and prints the frame periods)"""
import os, subprocess, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(wl, idx, K, L, start_at, frames):
    import torch
    from raytracer_amd import scenegen
    from raytracer_amd.executor import FramesInFlight
    W, H, depth, fn = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c3": (3840, 2160, 5, scenegen.scene_c2)}[wl]
    fs = fn().to_flat()
    pipe = FramesInFlight(fs, device=0, depth=K, lanes=L)
    outs = [torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]

    def run(n):
        for _ in range(n):
            k = pipe.next
            pipe.submit(W, H, outs[k].data_ptr(), None, max_depth=depth, tile_first=idx, tile_step=2)
        pipe.drain()
    run(3 * K)
    while time.time() < start_at:
        time.sleep(0.0005)
    t0 = time.perf_counter(); run(frames); dt = time.perf_counter() - t0
    print(f"worker {idx}: {dt / frames * 1e3:.3f} ms per half frame", flush=True)
    pipe.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "worker":
        worker(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]))
        sys.exit(0)
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    for queues in (4, 8):
        for K, L in ((1, 1), (1, 2), (2, 1), (2, 2), (4, 1)):
            env = dict(os.environ, GPU_MAX_HW_QUEUES=str(queues))
            for n_proc in (1, 2):
                start_at = time.time() + 25.0
                ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), wl, "worker", str(i), str(K), str(L), str(start_at), "60"],
                                       env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(n_proc)]
                outs = [p.communicate(timeout=180)[0].strip() for p in ps]
                if any(p.returncode for p in ps):
                    raise SystemExit(f"a worker failed: {outs}")
                ms = [float(o.split(":")[1].split()[0]) for o in outs]
                print(f"queues/process {queues}, frames in flight {K}, lanes {L}, processes {n_proc}: " + ", ".join(f"{m:.3f}" for m in ms) +
                      f" ms per half frame -> {max(ms):.3f} ms per whole frame" + (" (halves one after the other would take %.3f)" % (2 * ms[0]) if n_proc == 1 else ""), flush=True)
