# oracle/ref_hip.mk — TEST INFRASTRUCTURE ONLY: the link-time drop-in of INTEGRATION.md §1, performed.
#
# Compiles, from the sources WHERE THEY LIE under $(REF) (never copied), the reference's lib/src/phy MINUS the translation units
# that libsrslte_phy_hip.so replaces into _ref/hip/libsrslte_upper.a, and the reference's own test programs
# against it, linked with ../srslte-emane_amd/csrc/libsrslte_phy_hip.so in place of those translation units and of FFTW:
#   lib/test/phy/phy_dl_test.c, phch/test/{pdsch_test,pusch_test,pdsch_pdcch_file_test,pcfich_file_test,pbch_file_test,pmch_file_test}.c,
#   phch/test/pmch_test.c, fec/test/{turbodecoder_test,turbocoder_test}.c, dft/test/ofdm_test.c, utils/test/dft_test.c,
#   ch_estimation/test/{chest_test_dl,chest_test_ul}.c,
#   modem/test/{soft_demod_test,modem_test}.c, sync/test/sync_test.c (the reference's PSS / SSS search over this library's OFDM modulator and DFTs),
#   phch/test/{pbch_test,pcfich_test,pdcch_test,phich_test,phich_file_test}.c (control channels: the reference's own code over this library's
#   srslte_chest_dl_res_* / srslte_refsignal_cs_* / OFDM / estimator), phch/test/{prach_test,prach_test_multi}.c (the reference's PRACH generator and
#   detector over srslte_dft_*: 839- and 139-point sequences, 1536- to 24576-point transforms), phch/test/{pucch_test,pdcch_file_test}.c
# tests/test_gpu_dropin.py runs them on the GPU box with the reference's CTest arguments and asserts exit code 0.
# Nothing is stubbed: the same flags / force-include handling of the cmake-generated version.h as ref.mk.
#
# Output only into oracle/_ref/hip/ (git-ignored, travels with gpurun).
include ref.mk

HIPLIB   := ../srslte-emane_amd/csrc
HOUT     := $(OUT)/hip
HOBJ     := $(HOUT)/obj

# translation units served by libsrslte_phy_hip.so
REPLACED := dft/dft_fftw.c dft/ofdm.c dft/dft_precoding.c fec/turbodecoder.c fec/turbodecoder_gen.c fec/turbodecoder_sse.c fec/turbocoder.c \
            fec/cbsegm.c fec/tc_interl_lte.c fec/tc_interl_umts.c ch_estimation/chest_dl.c ch_estimation/refsignal_dl.c ch_estimation/chest_common.c \
            modem/demod_soft.c
# channel/: only the AWGN generator the tests use (the fading / delay / HST / RLF emulators are stimulus tools outside the path)
UP_DIRS  := agc ch_estimation common dft enb fec io mimo modem phch resampling scrambling sync ue utils
UP_C     := $(filter-out %viterbi37_neon.c $(addprefix $(RLIB)/src/phy/,$(REPLACED)),$(foreach d,$(UP_DIRS),$(wildcard $(RLIB)/src/phy/$(d)/*.c)) \
            $(RLIB)/src/phy/channel/ch_awgn.c $(RLIB)/src/phy/channel/gauss.c)
UP_OBJS  := $(patsubst $(RLIB)/src/phy/%.c,$(HOBJ)/%.o,$(UP_C)) $(HOBJ)/utils/random.o

TESTS    := lib/test/phy/phy_dl_test lib/src/phy/phch/test/pdsch_test lib/src/phy/phch/test/pusch_test lib/src/phy/phch/test/pdsch_pdcch_file_test \
            lib/src/phy/phch/test/pcfich_file_test lib/src/phy/phch/test/pbch_file_test lib/src/phy/phch/test/pmch_file_test \
            lib/src/phy/fec/test/turbodecoder_test lib/src/phy/fec/test/turbocoder_test lib/src/phy/dft/test/ofdm_test lib/src/phy/utils/test/dft_test \
            lib/src/phy/ch_estimation/test/chest_test_dl lib/src/phy/ch_estimation/test/chest_test_ul lib/src/phy/phch/test/pmch_test \
            lib/src/phy/modem/test/soft_demod_test lib/src/phy/modem/test/modem_test lib/src/phy/sync/test/sync_test \
            lib/src/phy/phch/test/pbch_test lib/src/phy/phch/test/pcfich_test lib/src/phy/phch/test/pdcch_test lib/src/phy/phch/test/phich_test \
            lib/src/phy/phch/test/phich_file_test lib/src/phy/phch/test/prach_test lib/src/phy/phch/test/prach_test_multi \
            lib/src/phy/phch/test/pdcch_file_test lib/src/phy/phch/test/pucch_test lib/src/phy/fec/test/rm_turbo_test
TEST_BIN := $(addprefix $(HOUT)/,$(notdir $(TESTS)))

.PHONY: ref_hip
ref_hip: $(TEST_BIN) $(HOUT)/dropin_log_test

# this repo's own caller: a registered log handler must receive the library's diagnostics (phy_logger.c:37-52)
$(HOUT)/dropin_log_test: dropin_log_test.c $(HOUT)/libsrslte_upper.a $(HIPLIB)/libsrslte_phy_hip.so
	gcc -std=c99 $(REF_FLAGS) $(FORCEINC) $< -o $@ $(HOUT)/libsrslte_upper.a -L$(HIPLIB) -lsrslte_phy_hip \
	    -Wl,-rpath,'$$ORIGIN/../../../srslte-emane_amd/csrc' -Wl,-rpath,/opt/rocm/lib -lstdc++ -lm -lpthread

# SCH_ON_DEVICE=1 (default): srslte_dlsch_decode2 / srslte_dlsch_decode come from libsrslte_phy_hip.so too - every code block of a transport
# block in one device call instead of one srslte_tdec_iteration round trip per block and pass. The reference's sch.c keeps everything else it
# defines (encoders, UL-SCH, UCI); its own two functions are compiled under other names (a -D on that one file, nothing is edited or stubbed).
SCH_ON_DEVICE ?= 1
ifeq ($(SCH_ON_DEVICE),1)
$(HOBJ)/phch/sch.o: EXTRA := -Dsrslte_dlsch_decode2=srslte_dlsch_decode2_on_host -Dsrslte_dlsch_decode=srslte_dlsch_decode_on_host
endif

$(HOBJ)/%.o: $(RLIB)/src/phy/%.c
	@mkdir -p $(dir $@)
	gcc -std=c99 $(if $(shell grep -l 'srslte/srslte\.h' $<),$(REF_FLAGS) $(FORCEINC),$(filter-out -DSRSLTE_SRSLTE_H,$(REF_FLAGS))) $(EXTRA) -c $< -o $@

$(HOBJ)/utils/random.o: $(RLIB)/src/phy/utils/random.cpp
	@mkdir -p $(dir $@)
	g++ -std=c++11 $(filter-out -DSRSLTE_SRSLTE_H,$(REF_FLAGS)) -c $< -o $@

# also on this file: a change of REPLACED alters the member list without touching any object
$(HOUT)/libsrslte_upper.a: $(UP_OBJS) ref_hip.mk
	@rm -f $@
	ar rcs $@ $(UP_OBJS)

# every test program includes the umbrella header
define TEST_RULE
$(HOUT)/$(notdir $(1)): $(REF)/$(1).c $(HOUT)/libsrslte_upper.a $(HIPLIB)/libsrslte_phy_hip.so
	gcc -std=c99 $$(REF_FLAGS) $$(FORCEINC) -I$(REF)/$(dir $(1)) $$< -o $$@ $(HOUT)/libsrslte_upper.a -L$(HIPLIB) -lsrslte_phy_hip \
	    -Wl,-rpath,'$$$$ORIGIN/../../../srslte-emane_amd/csrc' -Wl,-rpath,/opt/rocm/lib -lstdc++ -lm -lpthread
endef
$(foreach t,$(TESTS),$(eval $(call TEST_RULE,$(t))))

# phy_dl_test_sz: the same drop-in with ONE flag changed on ONE translation unit of the REFERENCE's own (not replaced) code -
# mimo/precoding.c with -fsigned-zeros, as libsrslte_ref_sz.so in ref.mk. With this image's gcc 11.4 the reference's -Ofast folds the two
# sign masks of its SIMD large-delay-CDD pre-decoders into one constant; `phy_dl_test -t 3` fails for that reason in every build made here,
# with or without this library. With the flag the test passes through this library's OFDM, estimator and decoder
# (tests/test_gpu_dropin.py::test_tm3_with_the_reference_predecoder_as_written).
$(HOBJ)/mimo/precoding_sz.o: $(RLIB)/src/phy/mimo/precoding.c
	@mkdir -p $(dir $@)
	gcc -std=c99 $(if $(shell grep -l 'srslte/srslte\.h' $<),$(REF_FLAGS) $(FORCEINC),$(filter-out -DSRSLTE_SRSLTE_H,$(REF_FLAGS))) -fsigned-zeros -c $< -o $@
$(HOUT)/libsrslte_upper_sz.a: $(UP_OBJS) $(HOBJ)/mimo/precoding_sz.o ref_hip.mk
	@rm -f $@
	ar rcs $@ $(filter-out $(HOBJ)/mimo/precoding.o,$(UP_OBJS)) $(HOBJ)/mimo/precoding_sz.o
$(HOUT)/phy_dl_test_sz: $(REF)/lib/test/phy/phy_dl_test.c $(HOUT)/libsrslte_upper_sz.a $(HIPLIB)/libsrslte_phy_hip.so
	gcc -std=c99 $(REF_FLAGS) $(FORCEINC) -I$(REF)/lib/test/phy/ $< -o $@ $(HOUT)/libsrslte_upper_sz.a -L$(HIPLIB) -lsrslte_phy_hip \
	    -Wl,-rpath,'$$ORIGIN/../../../srslte-emane_amd/csrc' -Wl,-rpath,/opt/rocm/lib -lstdc++ -lm -lpthread
ref_hip: $(HOUT)/phy_dl_test_sz
