"""Wall time of one frame at a time (device output, no gather) of a bench workload under the library's knobs of the
environment: python tools/single_frame_knobs.py c3 [lanes ...]   (HIPEXEC_* knobs are read when the executor is made;
MODE=1 in the environment renders with the cl_kernels.c semantics, one sample per pixel).
Prints per lane count the wall time per frame of the bound-sized first frames and of the later, exactly sized ones, with
the launches and the summed kernel time per class of a timed frame."""
import os, sys, time
os.environ.setdefault("HIPEXEC_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import _abi, scenegen
from raytracer_amd.executor import HipExecutor

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
lanes_list = [int(a) for a in sys.argv[2:]] or [1, 2]
mode = int(os.environ.get("MODE", "0"))
gen, kwargs, W, H, depth, desc = bench.WORKLOADS[wl]
fs = getattr(scenegen, gen)(**kwargs).to_flat()
dev = torch.device("cuda", 0)
out = torch.empty((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device=dev)
st = _abi.Stats()
for lanes in lanes_list:
    with HipExecutor(fs, device=0) as ex:
        ex.set_lanes(lanes)
        ts, redone = [], 0
        for i in range(9):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ex.render_device(W, H, out.data_ptr(), None, mode=mode, max_depth=depth, tone=True, stats=st, algo="bvh", time_kernels=(i == 8))
            ex.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
            redone += int(st.retraced)
        print("%s mode=%d lanes=%d retraced %d frames ms: %s | timed frame: device %.2f ms, kernel ms %s launches %s" % (
            wl, mode, lanes, redone, " ".join("%.2f" % t for t in ts[:8]), st.total_ms,
            [round(float(st.kernel_ms[k]), 2) for k in range(4)], [int(st.launches[k]) for k in range(4)]), flush=True)
