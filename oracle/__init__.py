"""oracle/ — CPU restatement (numpy) of the reference's NDT1-CTC train step.

TEST INFRASTRUCTURE ONLY. Nothing under llm_bci_amd/ may import this package: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker /
reported CPU baseline — never as the product path.

Parity status: PINNED. tests/golden/*.npz were generated in the build container by importing
the reference's own modules (tests/golden/make_golden.py, which needs /root/reference and is
not runnable on the GPU box); tests/test_oracle_golden.py checks every function here against
those fixtures.
"""
