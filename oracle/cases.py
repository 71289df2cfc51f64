"""oracle/cases.py — deterministic inputs shared by the pin script and the tests (TEST INFRASTRUCTURE)."""
import torch


def rng_inputs(N, R, Sc, Sf, seed=99):
    """The depth jitter [N,R²,Sc,1] and importance u [N·R²,Sf] both sides consume (CPU generator =>
    identical on every machine with the same torch build)."""
    g = torch.Generator().manual_seed(seed)
    jitter = torch.rand((N, R * R, Sc, 1), generator=g)
    u = torch.rand((N * R * R, Sf), generator=g)
    return jitter, u
