# timing-only ablations of k_blur_solve / k_update_matrices (results are wrong by design; only times matter).
# The sed patterns track the kernel sources; re-check them after editing the kernels.
set -e
cd funscript_flow_amd/csrc
cp kernels_farneback.hip /tmp/kf.orig; cp ffl_kernels.h /tmp/kh.orig
run() { rm -f kernels_farneback.o ffl_api.o kernels_post.o; make >/dev/null 2>&1; (cd ../..; timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-all | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],3), 'K5', round(k['k_blur_solve'],3), 'UM', round(k['k_update_matrices'],3))"); }
run baseline
sed -i 's/v\[j\] = \*reinterpret_cast<const float \*>(Mbb + (gy \* pitch + lane_byte));/v[j] = (float)(j + vtx);/' kernels_farneback.hip; run A_no_phaseV_loads
cp /tmp/kf.orig kernels_farneback.hip
sed -i 's/const float \*p = R1 + (size_t)y1 \* w + x1;/const float *p = R1 + (size_t)min(y, h - 2) * w + min(x, w - 2);/' ffl_kernels.h; run B_structured_gather
cp /tmp/kh.orig ffl_kernels.h
sed -i 's/\*reinterpret_cast<ffl_f2u \*>(Mo + c \* plane + o) = t;/if (t.x == 1.2345f) *reinterpret_cast<ffl_f2u *>(Mo + c * plane + o) = t;/' ffl_kernels.h; run C_no_M_stores
cp /tmp/kh.orig ffl_kernels.h
sed -i 's/ffl_box_block16(v, o);/for (int q = 0; q < TH; q++) o[q] = (double)v[q] + (double)v[q + 14];/; s/ffl_box_quarter<Q>(d, acc\[c0 + cc\]);/for (int q = 0; q < PX; q++) acc[c0 + cc][q] = d[q] + d[q + 14];/' kernels_farneback.hip; run D_no_box_adds
cp /tmp/kf.orig kernels_farneback.hip; cp /tmp/kh.orig ffl_kernels.h
# E: no R1 gathers at all (corners = constants)
sed -i 's|const ffl_f2u t = \*reinterpret_cast<const ffl_f2u \*>(p + c \* plane);.*|ffl_f2u t; t.x = t.y = (float)c;|; s|const ffl_f2u u = \*reinterpret_cast<const ffl_f2u \*>(p + c \* plane + w);.*|ffl_f2u u; u.x = u.y = fx;|' ffl_kernels.h; run E_no_R1_gathers
cp /tmp/kh.orig ffl_kernels.h
# F: no R0 loads
sed -i 's|const ffl_f2u t = \*reinterpret_cast<const ffl_f2u \*>(R0 + c \* plane + o);|ffl_f2u t; t.x = f0.x; t.y = f1.y;|' ffl_kernels.h; run F_no_R0_loads
cp /tmp/kh.orig ffl_kernels.h
rm -f kernels_farneback.o ffl_api.o kernels_post.o; make >/dev/null 2>&1
