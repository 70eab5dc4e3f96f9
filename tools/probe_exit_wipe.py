"""Probe: what does the device look like to the NEXT process right after a process that held a lot of device memory exits?
  python3 tools/probe_exit_wipe.py hold 100     allocate and touch 100 GB, exit
  python3 tools/probe_exit_wipe.py watch 12     poll rocm_smi for 12 s: VRAM used, SOC clock, socket power
Run as `hold` then `watch` in one shell line on the GPU box."""
import sys, os, time, ctypes as C


class Freqs(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


def smi():
    R = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
    assert R.rsmi_init(C.c_uint64(0)) == 0
    return R


def read(R):
    used, p, f = C.c_uint64(0), C.c_uint64(0), Freqs()
    R.rsmi_dev_memory_usage_get(C.c_uint32(0), C.c_int(0), C.byref(used))   # RSMI_MEM_TYPE_VRAM
    R.rsmi_dev_current_socket_power_get(C.c_uint32(0), C.byref(p))
    soc = -1.0
    if R.rsmi_dev_gpu_clk_freq_get(C.c_uint32(0), C.c_int(3), C.byref(f)) == 0 and f.current < 33:
        soc = f.frequency[f.current] / 1e6
    return used.value / 1e9, soc, p.value / 1e6


if sys.argv[1] == "hold":
    import torch
    gb = int(sys.argv[2])
    blocks = [torch.ones(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(gb)]
    torch.cuda.synchronize()
    print("holding %d GB: %s" % (gb, "VRAM used %.1f GB  socclk %.0f MHz  power %.0f W" % read(smi())), flush=True)
else:
    R = smi()
    hip = C.CDLL("libamdhip64.so")     # hipMemGetInfo next to rocm_smi: does the runtime's "free" see the pending wipe too?
    fr, tot = C.c_size_t(0), C.c_size_t(0)
    t0 = time.perf_counter()
    last = None
    while time.perf_counter() - t0 < float(sys.argv[2]):
        u, s, p = read(R)
        key = (round(u / 8), s > 600)
        if key != last:
            hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
            print("t = %6.2f s  VRAM used %6.1f GB  socclk %5.0f MHz  power %4.0f W   hipMemGetInfo free %6.1f of %.1f GB" %
                  (time.perf_counter() - t0, u, s, p, fr.value / 1e9, tot.value / 1e9), flush=True)
            last = key
        time.sleep(0.05)
