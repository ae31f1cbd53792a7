# same-box A/B of the bf16x3 / f16f8 / f16f6 Winograd conv (tools/bench_wino.py --f8), optional variant libraries (AB_LIBS="_x _y": built with
# MD_LIB_SUFFIX=_x MD_EXTRA_DEFINES="-DW8_HG=0" python -m meshdiffusion_amd.build; knobs: W8_HG, W8_NA, W8_PRO_FENCE, W8_PRO_SPLIT, W8_TAIL_RES,
# W8_FLUSH_LATE, timing-only W8_ABL), and the stamps of the
# f16f8 kernel (library built with MD_LIB_SUFFIX=_st MD_EXTRA_DEFINES=-DW8_STAMPS)
for rep in 1 2; do for v in "" $AB_LIBS; do echo "== lib$v"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --variants 0 --shapes 128:128:64:8,256:128:64:8,256:256:32:8 2>&1 | grep "A/B" | cut -c1-170; done; done
MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip_st.so python tools/bench_wino.py --f8 --stamps --variants 0 --shapes 128:128:64:8,256:128:64:8 2>&1 | grep stamps | cut -c1-420
