"""oracle/ — CPU restatement of the Next3D generator-forward hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import anything from this package; the product
(``next3d_amd``) never does and fails loudly when its HIP library is missing.

What it is: a from-scratch fp32 restatement (torch-CPU tensor ops + one small C file) of
every function on the path named by ``BASELINE.json:north_star`` — each function cites the
reference file:line it follows.  It takes a flat ``state_dict`` with the reference's
parameter names, so the very same tensors can be loaded into the real reference modules.

How it is pinned (see ``oracle/pin_against_reference.py`` and DESIGN.md §Oracle):
  * The reference ships NO tests / golden vectors for this path.
  * The reference's own Python IS importable in the build container, so the pin is:
    reference modules (``/root/reference``) run on CPU on seeded inputs  ==  this oracle,
    stage by stage, and the reference's outputs are committed as fixtures under
    ``tests/golden/`` together with the script that generated them.
  * The two third-party pieces the reference calls but does not ship
    (``pytorch3d...rasterize_meshes``, ``cv2.floodFill``) are restated from their
    published algorithms in ``oracle/raster_ref.c``: PARITY UNPINNED for those two
    boundaries (no PyTorch3D / OpenCV build exists in the container or the tree).
"""
