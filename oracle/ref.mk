# oracle/ref.mk — TEST INFRASTRUCTURE ONLY.
#
# Builds oracle/_ref/libsrslte_ref.so from the reference's own C sources WHERE THEY LIE under
# $(REF) (never copied into this repo), with the reference's release flags
# (CMakeLists.txt:355,372,392: -O3 -Ofast -funroll-loops -mfpmath=sse -mavx2 -mfma, LV_HAVE_*).
#
# What is NOT built, and why (see DESIGN.md "Oracle"):
#   * lib/src/phy/dft/dft_fftw.c needs <fftw3.h>/libfftw3f, absent from this image -> unbuildable.
#     Everything that reaches srslte_dft_* (ofdm.c, pss/sss FFT helpers, prach, fading, conv_fft) is dropped by
#     --gc-sections because the export map does not root it. dft/dft_precoding.c is compiled for its one pure
#     function, srslte_dft_precoding_valid_prb (which chest_ul.c calls); its FFTW-dependent functions are dropped too.
#   * Nothing is stubbed: no stand-in headers, libraries or generated files are written.
#     `srslte/srslte.h` pulls the cmake-generated `srslte/version.h` (version macros only); the
#     recipe pre-defines srslte.h's own include guard so that umbrella header contributes
#     nothing, and instead force-includes the very reference headers that umbrella lists
#     (derived from srslte.h itself at make time, minus version.h).
#
# Output only into oracle/_ref/ (git-ignored, NOT gpurun-ignored: it travels to the GPU box).

REF      ?= /root/reference
RLIB     := $(REF)/lib
OUT      := _ref
OBJ      := $(OUT)/obj

REF_DIRS := ch_estimation common fec mimo modem phch resampling scrambling utils sync
REF_C    := $(foreach d,$(REF_DIRS),$(wildcard $(RLIB)/src/phy/$(d)/*.c)) \
            $(RLIB)/src/phy/channel/ch_awgn.c $(RLIB)/src/phy/channel/gauss.c \
            $(RLIB)/src/phy/dft/dft_precoding.c
REF_CXX  := $(RLIB)/src/phy/utils/random.cpp
# NEON-only translation unit: not part of an x86 build of the reference
REF_C    := $(filter-out %viterbi37_neon.c,$(REF_C))

UMBRELLA := $(shell grep -o '"srslte/[a-z0-9_/]*\.h"' $(RLIB)/include/srslte/srslte.h 2>/dev/null | tr -d '"' | grep -v -e version.h -e config.h)
FORCEINC := -include complex.h -include math.h -include srslte/config.h $(foreach h,$(UMBRELLA),-include $(h))

REF_FLAGS := -D_GNU_SOURCE -O3 -fno-trapping-math -fno-math-errno -Ofast -funroll-loops \
             -mfpmath=sse -mavx2 -mfma -DLV_HAVE_SSE -DLV_HAVE_AVX -DLV_HAVE_AVX2 -DLV_HAVE_FMA \
             -DDISABLE_RF -fPIC -ffunction-sections -fdata-sections -w \
             -I$(RLIB)/include -I$(RLIB)/src/phy -DSRSLTE_SRSLTE_H

OBJS := $(patsubst $(RLIB)/src/phy/%.c,$(OBJ)/%.o,$(REF_C)) $(patsubst $(RLIB)/src/phy/%.cpp,$(OBJ)/%.o,$(REF_CXX))

.PHONY: ref all clean
ref: $(OUT)/libsrslte_ref.so

# the umbrella-header handling only for the translation units that include that header: force-including debug.h into the others
# would define DEBUG, which the Viterbi sources test with #ifdef to enable their metric dumps
$(OBJ)/%.o: $(RLIB)/src/phy/%.c
	@mkdir -p $(dir $@)
	gcc -std=c99 $(if $(shell grep -l 'srslte/srslte\.h' $<),$(REF_FLAGS) $(FORCEINC),$(filter-out -DSRSLTE_SRSLTE_H,$(REF_FLAGS))) -c $< -o $@

$(OBJ)/%.o: $(RLIB)/src/phy/%.cpp
	@mkdir -p $(dir $@)
	g++ -std=c++11 $(filter-out -DSRSLTE_SRSLTE_H,$(REF_FLAGS)) -c $< -o $@

$(OUT)/libsrslte_ref.so: $(OBJS) ref_exports.map
	g++ -shared -o $@ $(OBJS) -Wl,--gc-sections -Wl,--version-script=ref_exports.map -lm -lpthread
	@if nm -D --undefined-only $@ | grep -q 'srslte_\|fftw'; then echo "ERROR: unresolved reference symbols:"; nm -D --undefined-only $@ | grep 'srslte_\|fftw'; rm -f $@; exit 1; fi

# refdrv.c: this repo's own C driver over the reference library (the reference's callers' sequences from a frequency-domain grid on;
# see its header). Compiled against the reference headers, linked to libsrslte_ref.so.
ref: $(OUT)/librefdrv.so
$(OUT)/librefdrv.so: refdrv.c $(OUT)/libsrslte_ref.so
	gcc -std=c99 $(REF_FLAGS) $(FORCEINC) -shared -o $@ refdrv.c -L$(OUT) -lsrslte_ref -Wl,-rpath,'$$ORIGIN' -lm

# A second build that differs in ONE flag on ONE translation unit: mimo/precoding.c with -fsigned-zeros. The reference's flags contain
# -Ofast, which implies -fno-signed-zeros; with this image's gcc 11.4 the two sign masks {+0,-0,...} / {-0,+0,...} of the SIMD large-delay
# CDD pre-decoders (precoding.c:727-745,:929-947) then come out as ONE constant, the effective channel matrix gets two equal columns
# and TM3 reception fails in the build above (the reference's own `phy_dl_test -t 3` fails the same way). The scalar
# tail of the same functions is unaffected. This variant is what the source says; tests pin the two-layer modes on it and record the
# difference between the two builds (tests/test_oracle_vs_ref.py).
ref: $(OUT)/libsrslte_ref_sz.so
$(OBJ)/mimo/precoding_sz.o: $(RLIB)/src/phy/mimo/precoding.c
	@mkdir -p $(dir $@)
	gcc -std=c99 $(if $(shell grep -l 'srslte/srslte\.h' $<),$(REF_FLAGS) $(FORCEINC),$(filter-out -DSRSLTE_SRSLTE_H,$(REF_FLAGS))) -fsigned-zeros -c $< -o $@
$(OUT)/libsrslte_ref_sz.so: $(OBJS) $(OBJ)/mimo/precoding_sz.o ref_exports.map
	g++ -shared -o $@ $(filter-out $(OBJ)/mimo/precoding.o,$(OBJS)) $(OBJ)/mimo/precoding_sz.o -Wl,--gc-sections -Wl,--version-script=ref_exports.map -lm -lpthread
