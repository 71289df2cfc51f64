"""oracle/cases.py — deterministic inputs shared by the pin script and the tests (TEST INFRASTRUCTURE)."""
import torch


def rng_inputs(N, R, Sc, Sf, seed=99):
    """The depth jitter [N,R²,Sc,1] and importance u [N·R²,Sf] both sides consume (CPU generator =>
    identical on every machine with the same torch build)."""
    g = torch.Generator().manual_seed(seed)
    jitter = torch.rand((N, R * R, Sc, 1), generator=g)
    u = torch.rand((N * R * R, Sf), generator=g)
    return jitter, u


def block_inputs(tag, x_shape, img_shape, x_rms, img_rms):
    """Seeded inputs of a teacher-forced float16-block golden (oracle/pin_against_reference.py --fp16-blocks; tests/test_generator_gpu.py):
    x = float16(N(0,1) * per-channel rms of the reference's own activation at that point), img = float32 N(0,1) * per-channel rms — drawn
    from a CPU generator seeded by `tag`, so the reference (build container) and the HIP path (GPU box) get identical tensors without the
    inputs being stored."""
    import hashlib
    g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(tag.encode()).digest()[:7], 'little'))
    x = (torch.randn(x_shape, generator=g) * torch.as_tensor(x_rms, dtype=torch.float32).reshape(1, -1, 1, 1)).to(torch.float16)
    img = torch.randn(img_shape, generator=g) * torch.as_tensor(img_rms, dtype=torch.float32).reshape(1, -1, 1, 1)
    return x, img
