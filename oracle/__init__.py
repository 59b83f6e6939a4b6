"""CPU oracle for the UMHS volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker / timed baseline.
The product path (``unsupervised-hyperspectral-nerf_amd/``) never imports it
and fails loudly when the HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``ColourSystem`` / ``ClusterLookup`` / ``get_weights_spectral`` /
    ``blend_background*`` / ``UMHSField.get_outputs`` glue: PINNED against the
    reference's own code executed in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * nerfstudio==1.1.5 / nerfacc==0.5.2 pieces (hash encoding, MLP, NeRF/SH
    encodings, trunc_exp, contraction, packed transmittance, accumulate):
    those libraries are NOT in /root/reference and not installed, so they are
    restated from their published algorithm -- "parity unpinned" for exactly
    those functions (each is marked ``[upstream-recalled]`` in torch_ref.py).
"""
