"""Module-shaped adapters for the two THIRD-PARTY packages the reference's rasterisation step imports — `pytorch3d` and `cv2` — on
libn3d.so kernels (boundary B1 of SURVEY.md §8b): a pickled network executes the reference's own `Pytorch3dRasterizer.forward`
(volumetric_rendering/renderer.py:401-440), `fill_mouth` (:583-602) and `TriPlaneGenerator.__init__` (triplane_next3d.py:79-91) and
resolves `pytorch3d.*` / `cv2` at unpickle time; `next3d_amd.install_dropin(third_party=...)` registers these modules under those
names.  Only the call surface the generator-forward path uses exists; everything else raises.  No CPU arithmetic: the
rasteriser and the flood fill are the kernels of csrc/raster.hip (n3d_rasterize_meshes, n3d_flood_fill)."""
