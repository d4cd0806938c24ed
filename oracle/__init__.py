"""CPU restatement of the reference's SPR path -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py; never by openmeasure_amd."""
