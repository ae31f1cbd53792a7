# round 6: s_memtime + s_memrealtime stamps of the f16f8 Winograd conv -- round-5 kernel (_str5) vs HEAD (_st) on random data, and HEAD on all-zero
# data (the control: same cycles, higher clock).  Libraries: MD_LIB_SUFFIX=_st MD_EXTRA_DEFINES="-DW8_STAMPS -DW8_STAMPS_LATE" python -m meshdiffusion_amd.build
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for v in $ST_LIBS; do echo "== stamps lib$v"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --stamps --variants 0 --shapes 128:128:64:8,256:128:64:8 2>&1 | grep stamps | cut -c1-1200; done; done
echo "== stamps lib_st, all-zero activations and weights"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip_st.so python tools/bench_wino.py --f8 --stamps --zero-data --variants 0 --shapes 128:128:64:8,256:128:64:8 2>&1 | grep "stamps\|A/B" | cut -c1-1200
