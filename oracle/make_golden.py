"""Generates tests/golden/unet_forward_golden.npz by running the REFERENCE implementation
(/root/reference, imported -- never copied) on deterministic synthetic parameters and inputs.

Run in the build container only:   python oracle/make_golden.py
The fixture holds data only (probe coordinates, reference output values, per-channel statistics);
parameters and inputs are regenerated from seeds by oracle.unet_ref.synthetic_state_dict /
synthetic_input, so nothing of the reference's source travels.

Cases: variants {anatomix, anatomix-dev} x seeds {0, 1}; sizes 32^3 / 64^3
(6M) and 64^3 / 128^3 (dev), 128^3 (6M); plus the `layers` branch (taps 27,31,38,45,52,65 -> network.py:475-529) and
the encode_only early return for the 6M model.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from anatomix.model.network import Unet as RefUnet  # noqa: E402  (the reference itself)
from oracle import unet_ref as R  # noqa: E402

NPROBE = 4096
TAPS = [27, 31, 38, 45, 52, 65]       # pretraining/scripts/pretrain_anatomix.py:385


def probes(t: torch.Tensor, rs: np.random.RandomState):
    flat = t.reshape(-1)
    idx = rs.randint(0, flat.numel(), size=min(NPROBE, flat.numel())).astype(np.int64)
    return idx, flat[idx].numpy().astype(np.float32)


def chan_stats(t: torch.Tensor):
    c = t.shape[1]
    v = t.transpose(0, 1).reshape(c, -1).double()
    return np.stack([v.mean(1).numpy(), v.std(1, unbiased=False).numpy(), v.norm(dim=1).numpy()]).astype(np.float64)


def main():
    torch.set_num_threads(8)
    out = {}
    cases = [("anatomix", 0, 32, 1.0), ("anatomix", 1, 32, 1.0), ("anatomix", 0, 64, 1.0),
             ("anatomix", 0, 32, 2 ** 0.5), ("anatomix", 0, 128, 1.0),
             ("anatomix-dev", 0, 64, 1.0), ("anatomix-dev", 1, 64, 1.0), ("anatomix-dev", 0, 128, 1.0)]
    for variant, seed, size, gain in cases:
        kw = R.VARIANTS[variant]
        m = RefUnet(**kw).eval()
        sd = R.synthetic_state_dict(kw, seed, gain=gain)
        m.load_state_dict(sd, strict=True)
        x = R.synthetic_input(100 + seed, 1, (size,) * 3)
        with torch.no_grad():
            y = m(x)
        tag = f"{variant}|s{seed}|{size}|g{gain:.4f}"
        rs = np.random.RandomState(1000 + seed)
        idx, val = probes(y, rs)
        out[tag + "|idx"], out[tag + "|val"], out[tag + "|stats"] = idx, val, chan_stats(y)
        print(tag, "out std %.4f absmax %.4f" % (y.std().item(), y.abs().max().item()))
        if variant == "anatomix" and size == 32 and seed == 0 and gain == 1.0:
            out[tag + "|full"] = y.numpy().astype(np.float32)     # smallest legal cube, full tensor (2 MB)
            with torch.no_grad():
                y2, feats = m(x, TAPS, False)
                enc = m(x, TAPS[:2], True)
            assert torch.equal(y, y2) and len(enc) == 2
            for t, f in zip(TAPS, feats):
                i2, v2 = probes(f, np.random.RandomState(2000 + t))
                out[tag + f"|tap{t}|idx"], out[tag + f"|tap{t}|val"] = i2, v2
                out[tag + f"|tap{t}|shape"] = np.array(f.shape)
    path = os.path.join(ROOT, "tests", "golden", "unet_forward_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")
    taps_of_every_module_kind()


# One tap per module kind (network.py:475-529).  Two reference behaviours these pin down:
#   * a tap at an Upsample id is taken AFTER the concat with the skip (network.py:500-502 precede 504-516);
#   * the activation modules are built with inplace=True (network.py:188-196), so a tap at a NORM id aliases
#     the tensor the next ReLU overwrites: the caller sees post-activation values there.
KIND_TAPS = {"anatomix": (32, [0, 1, 2, 8, 9, 36, 37, 39, 58, 59, 64, 65]),
             "anatomix-dev": (64, [0, 1, 2, 9, 43, 44, 45, 79])}


def taps_of_every_module_kind():
    out = {}
    for variant, (size, taps) in KIND_TAPS.items():
        kw = R.VARIANTS[variant]
        m = RefUnet(**kw).eval()
        m.load_state_dict(R.synthetic_state_dict(kw, 0), strict=True)
        x = R.synthetic_input(100, 1, (size,) * 3)
        with torch.no_grad():
            y, feats = m(x, taps, False)
            enc = m(x, taps[:4], True)
        assert len(enc) == 4 and all(torch.equal(a, b) for a, b in zip(enc, feats))
        tag = f"{variant}|s0|{size}"
        out[tag + "|taps"] = np.array(taps)
        for t, f in zip(taps, feats):
            i2, v2 = probes(f, np.random.RandomState(3000 + t))
            out[tag + f"|tap{t}|idx"], out[tag + f"|tap{t}|val"] = i2, v2
            out[tag + f"|tap{t}|shape"] = np.array(f.shape)
            out[tag + f"|tap{t}|stats"] = np.array([f.double().mean().item(), f.double().norm().item(), f.min().item()])
    path = os.path.join(ROOT, "tests", "golden", "unet_taps_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
