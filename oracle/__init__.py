"""ORACLE -- CPU restatement of the reference's algorithm for the FreeFine hot path.

Test infrastructure only: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
the product package `freefine_amd` (which has no CPU fallback).  Every module cites the reference file:line it follows.
The reference is Python, so there is no `oracle/_ref` build; instead tools/gen_golden.py imports /root/reference in the
build container and commits small golden vectors under tests/golden/ that pin these restatements.
"""
