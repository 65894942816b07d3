# Builds libdr_mi355x.so (HIP kernels + engines + C ABI, gfx950 only) and the CPU oracle library.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := tandem_amd/csrc
LIB   := tandem_amd/libdr_mi355x.so
# -ffp-contract=off: the TSDF path is compared bit-for-bit with the C oracle (no FMA contraction on either side)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed
OBJS := $(CSRC)/dr_mvsnet.o $(CSRC)/dr_fusion.o $(CSRC)/dr_tracker.o
# the PARITY build: the same sources with -DDR_PARITY_HOOKS, i.e. with the superseded kernel generations (round-2 cost volume / prob /
# ray-caster, fused-skip convolutions, copy-engine result transfers) still selectable through their DR_* switches.  Test infrastructure:
# the -m gpu tests that compare generations load it (tests/conftest.py::parity_hooks); the product library does not contain them.
HLIB  := tandem_amd/libdr_mi355x_hooks.so
HOBJS := $(CSRC)/dr_mvsnet_hooks.o $(CSRC)/dr_fusion_hooks.o $(CSRC)/dr_tracker.o
# -Bsymbolic: both libraries export the same C ABI and may be loaded into one process; each binds its internal calls to itself
LDFLAGS := -shared -fPIC -Wl,-Bsymbolic

all: $(LIB) $(HLIB) oracle/libtsdf_oracle.so oracle/libtsdf_oracle_omp.so oracle/libtracker_oracle.so oracle/libtracker_oracle_left.so

MVS_DEPS := $(CSRC)/dr_mvsnet.hip $(CSRC)/conv_mfma.h $(CSRC)/conv_bf3.h $(CSRC)/conv_march.h $(CSRC)/conv_wino.h $(CSRC)/march_plan.h $(CSRC)/conv_tuned.h $(CSRC)/mvs_kernels.h $(CSRC)/tail_kernels.h $(CSRC)/fn_front.h $(CSRC)/fn_head3.h $(CSRC)/dr_common.h include/dr_mi355x.h
FUS_DEPS := $(CSRC)/dr_fusion.hip $(CSRC)/mesh_kernels.h $(CSRC)/mc_tables.h $(CSRC)/dr_common.h include/dr_mi355x.h
# the depth pipeline is held to a float tolerance, not to bit-exactness: let hipcc contract a*b+c into FMAs there (the
# vector-pipe kernels -- cost volume, prob -- are VALU-bound, and the reference's cuDNN/ATen kernels use FMAs too)
$(CSRC)/dr_mvsnet.o: $(MVS_DEPS)
	$(HIPCC) $(subst -ffp-contract=off,-ffp-contract=fast,$(HIPFLAGS)) -c $< -o $@
$(CSRC)/dr_mvsnet_hooks.o: $(MVS_DEPS)
	$(HIPCC) $(subst -ffp-contract=off,-ffp-contract=fast,$(HIPFLAGS)) -DDR_PARITY_HOOKS -c $< -o $@
$(CSRC)/dr_fusion.o: $(FUS_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/dr_fusion_hooks.o: $(FUS_DEPS)
	$(HIPCC) $(HIPFLAGS) -DDR_PARITY_HOOKS -c $< -o $@
$(CSRC)/dr_tracker.o: $(CSRC)/dr_tracker.hip $(CSRC)/dr_common.h include/dr_mi355x.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) $(LDFLAGS) $(OBJS) -o $@ -lpthread
$(HLIB): $(HOBJS)
	$(HIPCC) --offload-arch=$(ARCH) $(LDFLAGS) $(HOBJS) -o $@ -lpthread

oracle/libtsdf_oracle.so: oracle/tsdf_oracle.c tandem_amd/csrc/mc_tables.h
	gcc -O2 -std=c99 -fPIC -shared -ffp-contract=off -fno-fast-math $< -o $@ -lm

# the same restatement with its integration loop parallel over blocks (OpenMP): bench.py's multi-core CPU baseline only
oracle/libtsdf_oracle_omp.so: oracle/tsdf_oracle.c tandem_amd/csrc/mc_tables.h
	gcc -O2 -std=c99 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp $< -o $@ -lm

oracle/libtracker_oracle.so: oracle/tracker_oracle.c
	gcc -O2 -std=c99 -fPIC -shared -ffp-contract=off -fno-fast-math $< -o $@ -lm
# the same restatement with the other order of the hand-off's 3-term products, (a0 + a1) + a2: tests/test_ref_handoff.py caps what the order moves
oracle/libtracker_oracle_left.so: oracle/tracker_oracle.c
	gcc -O2 -std=c99 -fPIC -shared -ffp-contract=off -fno-fast-math -DTRK_SUM_LEFT $< -o $@ -lm

clean:
	rm -f $(OBJS) $(HOBJS) $(LIB) $(HLIB) oracle/libtsdf_oracle.so oracle/libtsdf_oracle_omp.so oracle/libtracker_oracle.so oracle/libtracker_oracle_left.so
.PHONY: all clean
