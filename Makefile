# Convenience targets (the driver uses __graft_entry__.py / pytest / bench.py directly).
PY ?= python

build:            ## hipcc --offload-arch=gfx950 for libqd.so, gcc for the oracle
	$(PY) -c 'import __graft_entry__ as g; g.build()'

test:             ## CPU suite: oracle vs golden vectors, host twin, C ABI, gloo
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity suite on an MI355X
	$(PY) -m pytest tests -q -m gpu

bench:            ## one JSON line: env steps/s, roofline, cpu_baseline
	$(PY) bench.py

sanitize:         ## ASan + UBSan over the CPU-runnable parts
	./tools/sanitize_cpu.sh

resources:        ## registers / LDS / scratch of every kernel
	$(PY) tools/kernel_resources.py

golden:           ## regenerate tests/golden/*.npz (needs /root/reference; build container only)
	$(PY) tests/golden/make_golden.py && $(PY) tests/golden/make_policy_golden.py && $(PY) tests/golden/make_stats_golden.py

.PHONY: build test test-gpu bench sanitize resources golden
