"""Scratch: shader clock and socket power while a workload loops (sysfs hwmon, 20 Hz), to tell a power-throttled kernel from a
stalled one.   python tools/clock_watch.py [forward|dev|mfma|idle]"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what = sys.argv[1] if len(sys.argv) > 1 else "forward"


def find(name):
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        out += glob.glob(os.path.join(card, "hwmon", "hwmon*", name))
    return out


freq, power = find("freq1_input"), find("power1_average") + find("power1_input")
print("sysfs:", freq[:1], power[:1], flush=True)
samples, stop = [], False


def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return -1


def poll():
    while not stop:
        f = rd(freq[0]) / 1e6 if freq else -1
        p = rd(power[0]) / 1e6 if power else -1
        samples.append((time.time(), f, p))
        time.sleep(0.05)


def smi():
    try:
        print(subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout[-900:], flush=True)
    except Exception as e:
        print("rocm-smi failed", e)


th = threading.Thread(target=poll, daemon=True)
th.start()
time.sleep(0.5)
if what == "mfma":
    subprocess.run(["tools/ubench/mfma_lds"], stdout=subprocess.DEVNULL)
elif what != "idle":
    import torch
    import anatomix_amd
    from oracle import unet_ref as R
    dev = torch.device("cuda:0")
    variant = "anatomix-dev" if what == "dev" else "anatomix"
    kw = R.VARIANTS[variant]
    m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
    if what == "dev":
        m.precision = "f16"
    x = R.synthetic_input(100, 4, (128, 128, 128)).to(dev)
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        samples.clear()
        t0 = time.time(); n = 0
        while time.time() - t0 < 6.0:
            for _ in range(50): m(x)
            torch.cuda.synchronize(); n += 50
        dt = time.time() - t0
    print(f"{variant}: {n * 4 / dt:.0f} vol/s over {dt:.1f} s", flush=True)
    smi_t = threading.Thread(target=smi); smi_t.start()
    with torch.no_grad():
        for _ in range(300): m(x)
        torch.cuda.synchronize()
    smi_t.join()
else:
    time.sleep(2)
stop = True
th.join()
fs = [s[1] for s in samples if s[1] > 0]; ps = [s[2] for s in samples if s[2] > 0]
if fs: print(f"sclk MHz: min {min(fs):.0f} median {sorted(fs)[len(fs)//2]:.0f} max {max(fs):.0f} ({len(fs)} samples)")
if ps: print(f"power W: min {min(ps):.0f} median {sorted(ps)[len(ps)//2]:.0f} max {max(ps):.0f}")
