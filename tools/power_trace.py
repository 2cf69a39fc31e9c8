"""Socket power and shader clock sampled (>= 10 Hz) while (a) the fused local transformer (local_pct6_kernel) and (b) a loop of nothing
but back-to-back v_mfma_f32_32x32x16_f16 on random operands (tools/experiments/mfma_power.hip, MODE 1) keep the GPU busy for ~3 s each,
plus an idle phase: the evidence for "the kernel runs on a power-managed clock" (DESIGN section 5).  Telemetry source: the amdgpu hwmon
files if visible (power1_average / power1_input in microwatts, freq1_input in Hz), else `rocm-smi --showpower --showclocks --json`
(or `amd-smi metric`).  Usage: python tools/power_trace.py > gpurun_out/power_trace.txt"""
import glob, json, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_hwmon():
    """hwmon directory of the GPU torch sees as cuda:0 (the node's other GPUs show up in sysfs too): matched by PCI address."""
    try:
        pr = torch.cuda.get_device_properties(0)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        dirs = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
    except Exception:
        dirs = []
    for d in dirs:
        p = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f))]
        if p and os.path.exists(os.path.join(d, "freq1_input")):
            return os.path.join(d, p[0]), os.path.join(d, "freq1_input")
    return None


def sample_hwmon(paths):
    with open(paths[0]) as f:
        w = float(f.read()) / 1e6
    with open(paths[1]) as f:
        mhz = float(f.read()) / 1e6
    return w, mhz


def sample_smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
    d = json.loads(out)
    card = d[sorted(d)[0]]
    w = next((float(v) for k, v in card.items() if "ower" in k and "W" in k and _num(v)), float("nan"))
    mhz = next((float(str(v).strip("()Mhz ")) for k, v in card.items() if k.startswith("sclk") and "clock speed" in k), float("nan"))
    return w, mhz


def _num(v):
    try:
        float(v)
        return True
    except Exception:
        return False


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.hw = find_hwmon()
        self.rows, self.tag, self.stop = [], "idle", False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            try:
                w, mhz = sample_hwmon(self.hw) if self.hw else sample_smi()
                self.rows.append((t, self.tag, w, mhz))
            except Exception as e:                       # keep sampling; report at the end
                self.rows.append((t, "error:" + repr(e)[:60], float("nan"), float("nan")))
            time.sleep(max(0.0, 0.02 - (time.perf_counter() - t)))


def main():
    dev = torch.device("cuda:0")
    import contextlib, io
    from macarons_amd import ops
    from macarons_amd.networks import SconeOcc
    from macarons_amd.networks.packing import pack_local_pct
    with contextlib.redirect_stdout(io.StringIO()):
        occ = SconeOcc().to(dev)
    blob = pack_local_pct(occ.local_transformers[0], 6)
    offs = torch.randn(16384 * 4, 16, 3, device=dev) * 0.05
    exe = "/tmp/mfma_power"
    if not os.path.exists(exe):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", os.path.join(ROOT, "tools", "experiments", "mfma_power.hip"),
                        "-o", exe], check=True)
    s = Sampler()
    s.start()
    time.sleep(1.5)
    s.tag = "local_pct6"
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(20):
            ops.local_pct_forward(offs, blob)
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    s.tag = "idle"
    print(f"local_pct6: {n} launches of 65536 queries in {dt:.2f} s = {dt / n / 4 * 1e3:.3f} ms per 16384 queries")
    time.sleep(1.5)
    s.tag = "mfma_only"
    t0 = time.perf_counter()
    out = ""
    while time.perf_counter() - t0 < 3.0:
        out = subprocess.run([exe], capture_output=True, text=True).stdout
    s.tag = "idle"
    time.sleep(1.0)
    s.stop = True
    s.join()
    print("mfma_power (last run):")
    print(out.strip())
    print(f"telemetry source: {'hwmon ' + s.hw[0] if s.hw else 'rocm-smi --showpower --showclocks --json'}; {len(s.rows)} samples")
    import collections
    by = collections.defaultdict(list)
    for t, tag, w, mhz in s.rows:
        by[tag].append((w, mhz))
    for tag, v in by.items():
        ws = [a for a, _ in v if a == a]
        fs = [b for _, b in v if b == b]
        if ws or fs:
            print(f"  {tag:12s} n={len(v):4d}  power W: mean {sum(ws) / max(len(ws), 1):7.1f}  max {max(ws) if ws else float('nan'):7.1f}   "
                  f"sclk MHz: mean {sum(fs) / max(len(fs), 1):7.1f}  min {min(fs) if fs else float('nan'):7.1f}  max {max(fs) if fs else float('nan'):7.1f}")
        else:
            print(f"  {tag}: {len(v)} samples, no readable values")
    print("samples (t s, phase, W, MHz), every 10th:")
    t00 = s.rows[0][0] if s.rows else 0
    for r in s.rows[::10]:
        print(f"  {r[0] - t00:6.2f} {r[1]:12s} {r[2]:8.1f} {r[3]:8.1f}")


if __name__ == "__main__":
    main()
