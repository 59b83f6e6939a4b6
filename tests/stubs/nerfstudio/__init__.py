"""Test double of the nerfstudio 1.1.5 API surface the UMHS plugin registers against  [upstream-recalled: nerfstudio is not
installable offline; signatures and behaviour restated from its published source].  Only tests put this directory on sys.path
(in a subprocess), to check that with a ``nerfstudio`` package importable ``umhs_config.umhs_method`` is a real
``MethodSpecification`` and that a Trainer-shaped loop can drive ``UMHSPipeline``.  It is not part of the product."""
