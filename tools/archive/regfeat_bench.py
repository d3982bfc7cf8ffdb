"""Times the registration feature post-processing kernels at registration sizes (256^3 image, grid_sp 2, disp_hw 1)
and prints achieved algorithmic bandwidth.  GPU only."""
import torch

from anatomix_amd.registration import MINDSSC, apply_avg_pool3d, correlate, smooth_merged_features


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    dev = torch.device("cuda:0")
    S = 256
    img = torch.rand(1, 1, S, S, S, device=dev)
    feats = torch.randn(1, 16, S, S, S, device=dev)
    mind = MINDSSC(img, 1, 2)
    vox = S ** 3
    t = timed(lambda: MINDSSC(img, 1, 2))
    print(f"MINDSSC(1,2) 256^3            {t:8.3f} ms   alg {(1 + 12) * 4 * vox / t / 1e6:8.1f} GB/s")
    t = timed(lambda: MINDSSC(img, 2, 2))
    print(f"MINDSSC(2,2) 256^3            {t:8.3f} ms   alg {(1 + 12) * 4 * vox / t / 1e6:8.1f} GB/s")
    t = timed(lambda: smooth_merged_features(mind, feats, 2, 0.1))
    print(f"cat + x0.1 + avg_pool(2)      {t:8.3f} ms   alg {(28 + 28 / 8) * 4 * vox / t / 1e6:8.1f} GB/s")
    sm = smooth_merged_features(mind, feats, 2, 0.1)
    mov = torch.roll(sm, (1, 0, -1), (2, 3, 4))
    g = S // 2
    t = timed(lambda: correlate(sm, mov, 1, 2, (S, S, S), 28))
    print(f"correlate hw=1 (28 ch, 128^3) {t:8.3f} ms   alg {(2 * 28 + 27) * 4 * g ** 3 / t / 1e6:8.1f} GB/s")
    x = torch.randn(1, 3, S, S, S, device=dev)
    t = timed(lambda: apply_avg_pool3d(x, 3, 3))
    print(f"apply_avg_pool3d(3, x3) 3ch   {t:8.3f} ms   alg {3 * 2 * 3 * 4 * vox / t / 1e6:8.1f} GB/s")
    # the same steps on stock torch ops, for scale
    def torch_path():
        a = torch.cat([mind, feats * 0.1], 1)
        return torch.nn.functional.avg_pool3d(a, 2, stride=2)
    t = timed(torch_path)
    print(f"torch: cat + x0.1 + avg_pool   {t:8.3f} ms")


if __name__ == "__main__":
    main()
