"""Probe: do the GPU's clocks / power / temperatures move with the SpMV's two speeds?  A sampler thread reads rocm_smi
(read-only: current sclk / mclk / fclk / socclk levels, socket power, junction and HBM temperature) every few ms while the
main thread runs (1) an idle second, (2) back-to-back SpMV bursts, (3) solves of 40 GMRES iterations with the in-library
profiler on (SpMV time per launch inside the Krylov loop), (4) bursts again.  Output: one line per phase window with the
mean / min / max of every sensor next to the SpMV time of that window."""
import sys, os, ctypes as C, time, threading, subprocess
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
KEEPALIVE = None
if os.environ.get("PROBE_KEEPALIVE"):   # built before anything touches the GPU
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(_HERE, "micro", "keepalive.hip"),
                    "-o", "/tmp/libkeepalive.so"], check=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields


class Freqs(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


class Sampler(threading.Thread):
    CLK = {"sclk": 0, "fclk": 1, "socclk": 3, "mclk": 4}
    TEMP = {"t_junction": 1, "t_mem": 2}

    def __init__(self, period=0.004):
        super().__init__(daemon=True)
        self.R = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
        rc = self.R.rsmi_init(C.c_uint64(0))
        if rc != 0:
            raise RuntimeError("rsmi_init %d" % rc)
        self.period, self.stop_flag, self.rows = period, False, []
        self.ok = {}

    def read(self):
        row = {"t": time.perf_counter()}
        f = Freqs()
        for name, k in self.CLK.items():
            if self.ok.get(name, True):
                rc = self.R.rsmi_dev_gpu_clk_freq_get(C.c_uint32(0), C.c_int(k), C.byref(f))
                self.ok[name] = rc == 0
                if rc == 0 and f.current < 33:
                    row[name] = f.frequency[f.current] / 1e6
        p = C.c_uint64(0)
        if self.ok.get("power", True):
            rc = self.R.rsmi_dev_current_socket_power_get(C.c_uint32(0), C.byref(p))
            self.ok["power"] = rc == 0
            if rc == 0:
                row["power_W"] = p.value / 1e6
        for name, k in self.TEMP.items():
            if self.ok.get(name, True):
                v = C.c_int64(0)
                rc = self.R.rsmi_dev_temp_metric_get(C.c_uint32(0), C.c_uint32(k), C.c_int(0), C.byref(v))
                self.ok[name] = rc == 0
                if rc == 0:
                    row[name] = v.value / 1e3
        return row

    def run(self):
        while not self.stop_flag:
            self.rows.append(self.read())
            time.sleep(self.period)

    def window(self, t0, t1):
        sel = [r for r in self.rows if t0 <= r["t"] <= t1]
        out = []
        for k in ("sclk", "fclk", "socclk", "mclk", "power_W", "t_junction", "t_mem"):
            v = [r[k] for r in sel if k in r]
            if v:
                out.append("%s %.0f [%.0f..%.0f]" % (k, sum(v) / len(v), min(v), max(v)))
        return "n=%d  " % len(sel) + "  ".join(out)


M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh, maxit=40, atol=0.0, rtol=0.0, quiet=True)
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
F = api.DeviceArray(6 * P.N)
P.assemble_system(wg_d, dwg_d, F, want_J=True)
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * P.N)); y = api.DeviceArray(6 * P.N)
sol = api.DeviceArray(6 * P.N)
L = api.lib()
L.DflProfileEnable.argtypes = [C.c_int]
L.DflProfileCollect.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.DflProfileCollect.restype = C.c_int
S = Sampler()
uid = C.c_uint64(0)
S.R.rsmi_dev_unique_id_get(C.c_uint32(0), C.byref(uid))
cap = C.c_uint64(0)
S.R.rsmi_dev_power_cap_get(C.c_uint32(0), C.c_uint32(0), C.byref(cap))
print("host %s  gpu unique id %016x  power cap %.0f W" % (os.uname().nodename, uid.value, cap.value / 1e6), flush=True)
print("sensors:", S.read(), flush=True)
S.start()
tm = api.Timer()


def phase(name, fn):
    t0 = time.perf_counter()
    r = fn()
    api.sync()
    t1 = time.perf_counter()
    print("%-34s %6.2f s  %-22s | %s" % (name, t1 - t0, r, S.window(t0, t1)), flush=True)


def idle():
    time.sleep(1.0)
    return ""


def bursts(n):
    def f():
        tm.start()
        for _ in range(n):
            P.matvec(x, y)
        tm.stop()
        return "SpMV %.4f ms" % (tm.ms() / n)
    return f


def nap(sec):
    def f():
        time.sleep(sec)
        return "sleep %.2f s" % sec
    return f


def solves(n):
    def f():
        L.DflProfileEnable(1)
        for _ in range(n):
            sol.zero()
            P.solve(sol, F)
        api.sync()
        tot, mn = C.c_double(0), C.c_double(0)
        cnt = L.DflProfileCollect(0, C.byref(tot), C.byref(mn))
        L.DflProfileEnable(0)
        return "in-loop SpMV %.4f ms" % (tot.value / max(cnt, 1))
    return f


def solves_keepalive(n, blocks, sleep_reps):
    inner = solves(n)
    def f():
        # ~0.6 us per iteration at sleep_reps = 1 (s_sleep 32 = 2048 cycles); bounded: ends by itself
        iters = int(n * 0.05 / (0.9e-6 * sleep_reps))
        assert KEEPALIVE.keepalive_start(C.c_long(iters), blocks, sleep_reps) == 0
        r = inner()
        t0 = time.perf_counter()
        assert KEEPALIVE.keepalive_wait() == 0
        return r + " (+%.2f s tail)" % (time.perf_counter() - t0)
    return f


if os.environ.get("PROBE_KEEPALIVE"):
    KEEPALIVE = C.CDLL("/tmp/libkeepalive.so")
    KEEPALIVE.keepalive_start.argtypes = [C.c_long, C.c_int, C.c_int]
phase("idle", idle)
for k in range(4):
    phase("burst of 500 SpMV", bursts(500))
if os.environ.get("PROBE_NAP_AFTER_FIRST"):   # first solve = work-space calibration (giant allocations and frees); then a short idle
    phase("1 solve (calibration)", solves(1))
    phase("nap", nap(float(os.environ["PROBE_NAP_AFTER_FIRST"])))
for k in range(6):
    phase("10 solves x 40 its (profiled)", solves(10))
if KEEPALIVE is not None:
    for blocks, reps in ((8, 1), (64, 1), (256, 1), (8, 8)):
        for k in range(2):
            phase("10 solves + keepalive %dx/%d" % (blocks, reps), solves_keepalive(10, blocks, reps))
        phase("10 solves x 40 its (profiled)", solves(10))
# what raises the SOC clock (1200 MHz during the first batch above, < 100 MHz later), and how short an idle drops it again?
host = np.random.default_rng(1).normal(size=6 * P.N)


def h2d():
    tmp = api.DeviceArray.from_numpy(host)
    api.sync()
    return "H2D copy of %.0f MB" % (host.nbytes / 1e6)


def d2h():
    y.numpy()
    return "D2H copy of %.0f MB" % (host.nbytes / 1e6)


for trigger, name in ((h2d, "H2D"), (d2h, "D2H")):
    phase("idle", idle)
    phase("10 solves x 40 its (profiled)", solves(10))
    phase(name, trigger)
    for k in range(3):
        phase("10 solves x 40 its (profiled)", solves(10))
    for sec in (0.02, 0.1, 0.5):
        phase("nap", nap(sec))
        for k in range(2):
            phase("10 solves x 40 its (profiled)", solves(10))
S.stop_flag = True
