"""Output side of the video scripts: float frames -> uint8 -> tiled grid image.

`layout_grid` keeps the signature and result of the reference helper both video scripts define (gen_videos_next3d.py:35-49,
reenact_avatar_next3d.py:56-70).  The float -> uint8 conversion runs in libn3d.so (`n3d_to_uint8`, the same kernel bench.py
uses for gen_samples_next3d.py:201), so what crosses PCIe for a 2x2 grid of 512² frames is 3 MB of uint8 instead of 12.6 MB
of fp32; the tiling itself is data movement (reshape / permute), which stays in torch like every other copy of the path.
"""
import torch

from . import _lib


def to_uint8(img):
    """(img * 127.5 + 128).clamp(0, 255).to(uint8) on the device (gen_samples_next3d.py:201); layout preserved."""
    _lib.require_device(img)
    img = img.to(torch.float32).contiguous()
    out = torch.empty(img.shape, dtype=torch.uint8, device=img.device)
    _lib.check(_lib.lib().n3d_to_uint8(_lib.ptr(img), _lib.ptr(out), img.numel(), _lib.stream()))
    return out


def layout_grid(img, grid_w=None, grid_h=1, float_to_uint8=True, chw_to_hwc=True, to_numpy=True):
    """[B,C,H,W] -> one [grid_h*H, grid_w*W, C] image, frame b at row b // grid_w, column b % grid_w."""
    batch_size, channels, img_h, img_w = img.shape
    if grid_w is None:
        grid_w = batch_size // grid_h
    assert batch_size == grid_w * grid_h
    if float_to_uint8:
        img = to_uint8(img)
    img = img.reshape(grid_h, grid_w, channels, img_h, img_w)
    img = img.permute(2, 0, 3, 1, 4)
    img = img.reshape(channels, grid_h * img_h, grid_w * img_w)
    if chw_to_hwc:
        img = img.permute(1, 2, 0)
    if to_numpy:
        img = img.cpu().numpy()
    return img
