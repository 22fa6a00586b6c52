"""CPU oracle for the GAN-vocoder hot path  --  TEST INFRASTRUCTURE ONLY.

Nothing in ``parallelwavegan_amd`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / the timed CPU baseline, never as the product path.

Contents
--------
``torch_cpu``   functional restatement of the reference's L1 modules
                (models/layers/losses) as plain ``torch`` CPU calls.  The
                reference is 100 % Python over ATen ops, so the
                faithful CPU restatement *is* a sequence of ATen CPU calls.
``slaney_mel``  numpy restatement of ``librosa.filters.mel`` (librosa is not
                vendored by the reference and not installed here; parity of the
                basis values themselves is therefore UNPINNED, see header there).
``c/``          plain-C restatement of the ATen primitives the path bottoms out
                in (conv1d / conv_transpose1d / avg_pool1d / PQMF FIR / DFT
                magnitude), used to check the torch-level oracle and the HIP
                kernels independently of any torch build.
``ref_shim``    import shim that lets the *unmodified* reference under
                /root/reference run in the build container (it does not exist on
                the GPU box).  Used only by ``tests/golden/make_golden.py`` and
                by CPU tests that skip when /root/reference is absent.

Parity status: pinned against outputs of the reference itself, run in the build
container through ``ref_shim`` (fixtures in tests/golden/, generator committed).
The reference ships no golden vectors of its own.
"""
