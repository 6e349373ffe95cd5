# timing-only ablations of k_update_matrices (results are wrong by design; only times matter)
set -e
cd funscript_flow_amd/csrc
cp kernels_farneback.hip /tmp/kf.orig; cp ffl_kernels.h /tmp/kh.orig
run() { rm -f kernels_farneback.o ffl_api.o kernels_post.o; make >/dev/null 2>&1; (cd ../..; timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-all | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],3), 'K5', round(k['k_blur_solve'],3), 'UM', round(k['k_update_matrices'],3))"); }
run baseline
# A: no R1 gathers (the bilinear block sees constants)
sed -i 's/const ffl_f2u t = \*reinterpret_cast<const ffl_f2u \*>(p + c \* plane);      \/\/ (x1, y1), (x1+1, y1)/ffl_f2u t; t.x = fx; t.y = fy;/; s/const ffl_f2u u = \*reinterpret_cast<const ffl_f2u \*>(p + c \* plane + w);  \/\/ (x1, y1+1), (x1+1, y1+1)/ffl_f2u u; u.x = fy; u.y = fx;/' ffl_kernels.h; run A_no_R1_gathers
cp /tmp/kh.orig ffl_kernels.h
# B: no R0 loads
sed -i 's/const ffl_f2u t = \*reinterpret_cast<const ffl_f2u \*>(R0 + c \* plane + o);/ffl_f2u t; t.x = f0.x; t.y = f1.y;/' ffl_kernels.h; run B_no_R0_loads
cp /tmp/kh.orig ffl_kernels.h
# C: no M stores
sed -i 's/\*reinterpret_cast<ffl_f2u \*>(Mo + c \* plane + o) = t;/if (t.x == 1.2345f) *reinterpret_cast<ffl_f2u *>(Mo + c * plane + o) = t;/' ffl_kernels.h; run C_no_M_stores
cp /tmp/kh.orig ffl_kernels.h
# D: no upsample reads (flow from constants)
sed -i 's/const float2 p00 = r0p\[xa0\], p01 = r0p\[xa1\], p10 = r1p\[xa0\], p11 = r1p\[xa1\];/const float2 p00 = make_float2(b0, b1), p01 = p00, p10 = p00, p11 = p00;/; s/const float2 p00 = r0p\[xb0\], p01 = r0p\[xb1\], p10 = r1p\[xb0\], p11 = r1p\[xb1\];/const float2 p00 = make_float2(b1, b0), p01 = p00, p10 = p00, p11 = p00;/' kernels_farneback.hip; run D_no_prev_flow_reads
cp /tmp/kf.orig kernels_farneback.hip; cp /tmp/kh.orig ffl_kernels.h
