"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the DualVGR reasoning hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the timed CPU
baseline -- never as the implementation that is measured or shipped.  The product path
(``dualvgr_videoqa_amd``) never imports this package and raises when its HIP library is missing.

Parity status: PINNED.  The reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself, imported
in the build container from /root/reference by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (inputs, weights, all 7 outputs, losses and gradients).
``tests/test_oracle_golden.py`` checks every function here against those vectors.
"""
