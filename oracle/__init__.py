"""CPU oracle for the arreau sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU) restatement of
the reference's reverse-diffusion step.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed baseline.  The product path
(``arreau_amd``) never imports it and fails loudly when the HIP library is
missing.

Pinning status (see DESIGN.md "Oracle"):

* pinned by fixtures generated from the reference's own importable modules
  (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``): ``radius_graph_pbc``,
  ``lattice_from_params``, ``matrix_to_params``, ``frac_to_cart_coords``,
  VE/VP schedules and their reverse updates, ``D3PM`` buffers /
  ``q_posterior_logits`` / ``reverse``, ``GaussianFourierProjection``,
  ``invariant_attr_r3s2_fiber_bundle``, ``PolynomialFeatures``,
  ``PolynomialCutoff``, the sphere maps, ``ConvNext`` and ``uniform_grid_s2``.
* restated from source only (their modules need torch_geometric /
  pytorch_lightning, which are not installed here): ``FiberBundleConv``,
  ``SEnInvariantAttributes``, ``PositionOrientationGraph``,
  ``PonitaFiberBundle.forward``, ``DiffusionLoss.predict_scores`` / ``sample``.
  For those rows parity is "unpinned" beyond the piecewise fixtures above.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
