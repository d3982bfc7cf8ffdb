"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the anatomix UNet hot path.

Nothing under ``anatomix_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / reported baseline, never as the product path.
"""
