"""Packed driving-mesh sequences for the reenactment loop.

The reference re-parses one Wavefront `.obj` (5,023 `v` lines) plus one `_kpt2d.txt` (68 landmarks) in Python for every
frame (reenact_avatar_next3d.py:139-148, gen_samples_next3d.py:165-179): ~20 ms of text parsing per frame, i.e. more than
the 3.4 ms the generator forward takes on one MI355X.  `pack_sequence` converts a frame list ONCE into a flat little-endian
file — 32-byte header + float32 [T, 5091, 3] (vertices followed by landmarks, exactly the `v` tensor `synthesis` takes) —
and `MeshSequence` memory-maps it and streams batches to the GPU through two pinned staging buffers on a copy stream, so
the host side costs one memcpy per batch and overlaps the previous batch's forward.

File layout: magic b'N3DMESH1', uint32 T, uint32 V (5023), uint32 L (68), uint32 reserved (0), 8 bytes zero,
then T * (V + L) * 3 float32.
"""
import os
import struct

import numpy as np
import torch

from . import mesh

MAGIC = b'N3DMESH1'
HEADER = struct.Struct('<8sIIII8x')


def pack_sequence(obj_paths, landmark_paths, out_path):
    """Parse every (`.obj`, `_kpt2d.txt`) pair once and write the packed file; returns the [T, V+L, 3] array."""
    assert len(obj_paths) == len(landmark_paths) and len(obj_paths) > 0
    frames = []
    for o, k in zip(obj_paths, landmark_paths):
        v = mesh.parse_obj_vertices(o)[0].numpy()
        l = mesh.parse_landmarks(k)[0].numpy()
        frames.append(np.concatenate([v, l], 0).astype('<f4'))
    shapes = {f.shape for f in frames}
    if len(shapes) != 1:
        raise ValueError(f'frames disagree on the vertex / landmark count: {sorted(shapes)}')
    arr = np.stack(frames, 0)
    V, L = mesh.parse_obj_vertices(obj_paths[0]).shape[1], mesh.parse_landmarks(landmark_paths[0]).shape[1]
    tmp = out_path + '.tmp'
    with open(tmp, 'wb') as fh:
        fh.write(HEADER.pack(MAGIC, arr.shape[0], V, L, 0))
        fh.write(arr.tobytes())
    os.replace(tmp, out_path)
    return arr


class MeshSequence:
    """Memory-mapped packed sequence.  `seq[i]` -> [V+L, 3] float32 numpy view; `batches(n, device)` yields [n, V+L, 3]
    device tensors, the upload of batch k+1 overlapping whatever the caller does with batch k."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            head = fh.read(HEADER.size)
        if len(head) != HEADER.size:
            raise ValueError(f'{path}: truncated header')
        magic, T, V, L, _ = HEADER.unpack(head)
        if magic != MAGIC:
            raise ValueError(f'{path}: not a packed mesh sequence (magic {magic!r})')
        expect = HEADER.size + T * (V + L) * 3 * 4
        if os.path.getsize(path) != expect:
            raise ValueError(f'{path}: size {os.path.getsize(path)} != {expect} implied by the header')
        self.T, self.V, self.L = T, V, L
        self.data = np.memmap(path, dtype='<f4', mode='r', offset=HEADER.size, shape=(T, V + L, 3))

    def __len__(self):
        return self.T

    def __getitem__(self, i):
        return self.data[i]

    def batches(self, n, device, start=0, stop=None, drop_last=False):
        stop = self.T if stop is None else min(stop, self.T)
        device = torch.device(device)
        if device.type != 'cuda':
            for i in range(start, stop, n):
                if drop_last and i + n > stop:
                    return
                yield torch.from_numpy(np.array(self.data[i:min(i + n, stop)]))      # a writable copy, not a view of the read-only map
            return
        copy_stream = torch.cuda.Stream(device=device)
        stage = [torch.empty(n, self.V + self.L, 3, dtype=torch.float32).pin_memory() for _ in range(2)]
        free = [None, None]                          # event after which staging buffer k may be overwritten

        def upload(i, k):
            m = min(n, stop - i)
            if free[k] is not None:
                free[k].synchronize()
            stage[k][:m].copy_(torch.from_numpy(np.ascontiguousarray(self.data[i:i + m])))
            with torch.cuda.stream(copy_stream):
                dev = stage[k][:m].to(device, non_blocking=True)
                free[k] = copy_stream.record_event()
            return dev, free[k]

        idx = list(range(start, stop, n))
        if drop_last and idx and idx[-1] + n > stop:
            idx.pop()
        pending = upload(idx[0], 0) if idx else None
        for j, i in enumerate(idx):
            dev, ev = pending
            pending = upload(idx[j + 1], (j + 1) & 1) if j + 1 < len(idx) else None
            torch.cuda.current_stream(device).wait_event(ev)
            dev.record_stream(torch.cuda.current_stream(device))
            yield dev
