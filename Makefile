# Same targets as the reference's Makefile (Makefile:14-20 there): `make` builds the executable,
# `make check` compares ./final_state.dat and ./av_vels.dat with the reference results.
EXE=d2q9-bgk

FINAL_STATE_FILE=./final_state.dat
AV_VELS_FILE=./av_vels.dat
REF_FINAL_STATE_FILE=tests/golden/check/128x128.final_state.dat.gz
REF_AV_VELS_FILE=tests/golden/check/128x128.av_vels.dat.gz

all: $(EXE)

$(EXE): mpilattice-boltzmann_amd/csrc/*.hip mpilattice-boltzmann_amd/csrc/*.cpp mpilattice-boltzmann_amd/csrc/*.c mpilattice-boltzmann_amd/csrc/*.h mpilattice-boltzmann_amd/csrc/kernels/*.h include/*.h
	python3 -c "import __graft_entry__ as g; g.build()"
	ln -sf mpilattice-boltzmann_amd/bin/d2q9-bgk $(EXE)

check:
	python3 check/check.py --ref-av-vels-file=$(REF_AV_VELS_FILE) --ref-final-state-file=$(REF_FINAL_STATE_FILE) --av-vels-file=$(AV_VELS_FILE) --final-state-file=$(FINAL_STATE_FILE)

test:
	python3 -m pytest tests -x -q -m "not gpu"

.PHONY: all check clean test

clean:
	rm -f $(EXE) final_state.dat av_vels.dat
	rm -rf mpilattice-boltzmann_amd/lib mpilattice-boltzmann_amd/bin
	$(MAKE) -C oracle clean
