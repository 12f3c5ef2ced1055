"""CPU oracle of the MI355X AutoRound hot path: TEST INFRASTRUCTURE ONLY (see ar_oracle.c / oracle.py / torch_ref.py headers).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
