"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the Wan 2.1/2.2 denoise hot path.  Nothing under ``oracle/`` may be
imported by the product package ``wan2gp_amd``; only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` use it, as the checker.
"""
