"""Host-side logic of the sliding-window dispatcher: which heads behind nn.Sequential(Unet, head) may be applied ONCE after the window
averaging (anatomix_amd/registration/sliding_window.py _pointwise_affine).  Only a provable per-voxel affine map commutes with the
averaging; a module with its own forward() can hide a functional nonlinearity that no child module reveals (advisor, round 4)."""
import torch
from torch import nn

from anatomix_amd.registration.sliding_window import _pointwise_affine


class SoftmaxInForward(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv3d(16, 5, 1)

    def forward(self, x):
        return torch.softmax(self.conv(x), 1)


class PlainWrapper(nn.Module):              # affine in fact, but its forward() is unknown code: not accepted structurally
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv3d(16, 5, 1)

    def forward(self, x):
        return self.conv(x)


class SequentialWithOwnForward(nn.Sequential):
    def forward(self, x):
        return torch.sigmoid(super().forward(x))


def test_pure_compositions_of_1x1x1_convs_are_affine():
    assert _pointwise_affine(nn.Conv3d(16, 5, 1))
    assert _pointwise_affine(nn.Sequential(nn.Sequential(nn.Conv3d(16, 8, 1)), nn.Identity(), nn.Conv3d(8, 3, 1, bias=False)))
    assert _pointwise_affine(nn.Sequential(nn.Conv3d(16, 5, 1), nn.Dropout3d(0.5)).eval())


def test_heads_that_are_not_provably_affine_keep_the_generic_loop():
    assert not _pointwise_affine(SoftmaxInForward())
    assert not _pointwise_affine(nn.Sequential(SoftmaxInForward()))
    assert not _pointwise_affine(PlainWrapper())
    assert not _pointwise_affine(SequentialWithOwnForward(nn.Conv3d(16, 5, 1)))
    assert not _pointwise_affine(nn.Sequential(nn.Conv3d(16, 5, 1), nn.Softmax(dim=1)))
    assert not _pointwise_affine(nn.Sequential(nn.Conv3d(16, 5, 3, padding=1)))           # not per voxel
    assert not _pointwise_affine(nn.Sequential(nn.Conv3d(16, 5, 1), nn.Dropout3d(0.5)).train())
    assert not _pointwise_affine(nn.Sequential(nn.Conv3d(16, 5, 1), nn.ReLU()))
