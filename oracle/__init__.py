"""CPU oracle for the multispectral index path (TEST INFRASTRUCTURE ONLY).

Nothing in the shipped package imports this directory.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.
"""
