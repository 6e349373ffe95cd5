# timing-only ablations of k_blur_solve / k_update_matrices (results are wrong by design; only times matter).
# The sed patterns track the kernel sources; re-check them after editing the kernels.
set -e
cd funscript_flow_amd/csrc
cp kernels_farneback.hip /tmp/kf.orig; cp ffl_kernels.h /tmp/kh.orig
run() { rm -f kernels_farneback.o ffl_api.o kernels_post.o; make >/dev/null 2>&1; (cd ../..; timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-all | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],3), 'K5', round(k['k_blur_solve'],3), 'UM', round(k['k_update_matrices'],3))"); }
run baseline
sed -i 's/v\[j\] = row\[lane_off\];/v[j] = (float)(j + tx);/' kernels_farneback.hip; run A_no_phaseV_loads
cp /tmp/kf.orig kernels_farneback.hip
sed -i 's/const float \*p = R1 + (size_t)y1 \* w + x1;/const float *p = R1 + (size_t)min(y, h - 2) * w + min(x, w - 2);/' ffl_kernels.h; run B_structured_gather
cp /tmp/kh.orig ffl_kernels.h
sed -i 's/\*reinterpret_cast<ffl_f2u \*>(Mo + c \* plane + o) = t;/if (t.x == 1.2345f) *reinterpret_cast<ffl_f2u *>(Mo + c * plane + o) = t;/' ffl_kernels.h; run C_no_M_stores
cp /tmp/kh.orig ffl_kernels.h
sed -i 's/ffl_box15_run<TH>(v, o);/for (int q = 0; q < TH; q++) o[q] = (double)v[q] + (double)v[q + 14];/; s/ffl_box15_run<PX>(d, acc\[c0 + cc\]);/for (int q = 0; q < PX; q++) acc[c0 + cc][q] = d[q] + d[q + 14];/' kernels_farneback.hip; run D_no_box_adds
cp /tmp/kf.orig kernels_farneback.hip; cp /tmp/kh.orig ffl_kernels.h
