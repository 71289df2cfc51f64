"""Host-side mirror of the reference's `torch_utils` operator layer (the B1 drop-in boundary):
same module paths, function names, argument meaning and error behaviour — backed by libn3d.so only."""
