# Top-level convenience targets (the Python driver hooks call the same commands: __graft_entry__.build()).
#   make            libnmi_hip.so (hipcc, gfx950) + CPU oracle + C++ demo
#   make test       CPU test tier;  make test-gpu  GPU tier (needs an MI355X)
PY ?= python3

all: lib oracle demo

lib:
	$(PY) -m orbslam2_nmi_amd.build

oracle:
	$(MAKE) -C oracle

demo: lib
	$(MAKE) -C examples

test: all
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: all
	$(PY) -m pytest tests -x -q -m gpu

clean:
	rm -f orbslam2_nmi_amd/lib/libnmi_hip.so oracle/libnmi_oracle.so examples/relocalize_demo

.PHONY: all lib oracle demo test test-gpu clean
