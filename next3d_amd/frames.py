"""Output side of the video scripts: float frames -> one tiled uint8 canvas.

`layout_grid` answers the same call as the helper both video scripts define for themselves (gen_videos_next3d.py:35-49,
reenact_avatar_next3d.py:56-70: same arguments, same result) — the unchanged scripts keep using their own; this one serves callers of
the package (tests, the benchmark's uint8 gather).  The default call (float frames on the device -> uint8) is ONE libn3d.so launch,
`n3d_layout_grid_u8`: conversion, tiling and CHW -> HWC in a single pass, so a 2x2 grid of 512² frames is read once (12.6 MB of
float32) and leaves the device as 3 MB of uint8.  Without the conversion (`float_to_uint8=False`) tiling is pure data movement: every
frame is copied into its tile of a pre-allocated canvas.
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


def _tile(frames, cols, rows):
    """[B,C,H,W] (any dtype / device) -> [C, rows*H, cols*W]: frame k goes to tile (k // cols, k % cols)."""
    count, depth, th, tw = frames.shape
    canvas = frames.new_empty((depth, rows * th, cols * tw))
    for k in range(count):
        r, c = divmod(k, cols)
        canvas[:, r * th:(r + 1) * th, c * tw:(c + 1) * tw] = frames[k]
    return canvas


def layout_grid(img, grid_w=None, grid_h=1, float_to_uint8=True, chw_to_hwc=True, to_numpy=True):
    """[B,C,H,W] -> one [grid_h*H, grid_w*W, C] image (or [C, grid_h*H, grid_w*W] with chw_to_hwc=False), frame b at tile row
    b // grid_w, tile column b % grid_w."""
    count = img.shape[0]
    cols = count // grid_h if grid_w is None else grid_w
    assert cols * grid_h == count
    if float_to_uint8:
        _lib.require_device(img)                                  # the conversion is a libn3d.so kernel: no CPU fallback
        src = img.to(torch.float32).contiguous()
        _, depth, th, tw = src.shape
        if tw % 4 == 0:                                           # conversion + tiling + CHW -> HWC in one launch
            shape = (grid_h * th, cols * tw, depth) if chw_to_hwc else (depth, grid_h * th, cols * tw)
            canvas = torch.empty(shape, dtype=torch.uint8, device=src.device)
            _lib.check(_lib.lib().n3d_layout_grid_u8(_lib.ptr(src), _lib.ptr(canvas), count, depth, th, tw, cols, grid_h,
                                                     1 if chw_to_hwc else 0, _lib.stream()))
            return canvas.cpu().numpy() if to_numpy else canvas
        img = to_uint8(src)
    canvas = _tile(img, cols, grid_h)
    if chw_to_hwc:
        canvas = canvas.movedim(0, 2)
    return canvas.cpu().numpy() if to_numpy else canvas
