/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the two bit-exact (integer / byte-copy) pieces of the hot path, used
 * by tests/ as a second, framework-free checker next to oracle/ref_model.py:
 *   - get_patch            ACT/models/utils.py:37-51 (= STH/models/utils.py:44-58)
 *   - TemporalShift.shift  STH/ops/temporal_shift.py:28-46
 * Pinned against tests/golden/g1..g3 (vectors produced by the real reference) in
 * tests/test_oracle_golden.py::test_c_oracle_*.
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* patch_coordinate = torch.floor(action_sequence * (image_size - patch_size)).int()
 * fp32 multiply by the int promoted to fp32, floor, truncate (utils.py:42). */
void ref_patch_coords(const float* action_yx, int n, int image_size, int patch_size, int32_t* coords) {
    const float span = (float)(image_size - patch_size);
    for (int i = 0; i < 2 * n; ++i) {
        volatile float prod = action_yx[i] * span; /* volatile: force the fp32 rounding of the product */
        coords[i] = (int32_t)floorf(prod);
    }
}

/* images [n,c,h,w] -> out [n,c,p,p]; rows from coords[i][0], columns from coords[i][1] (utils.py:44-47) */
void ref_get_patch(const float* images, int n, int c, int h, int w, const float* action_yx, int p, float* out) {
    for (int i = 0; i < n; ++i) {
        int32_t yx[2];
        ref_patch_coords(action_yx + 2 * i, 1, h, p, yx);
        for (int ch = 0; ch < c; ++ch)
            for (int y = 0; y < p; ++y)
                memcpy(out + (((size_t)i * c + ch) * p + y) * p,
                       images + (((size_t)i * c + ch) * h + (yx[0] + y)) * w + yx[1], (size_t)p * sizeof(float));
    }
}

/* x [n_batch*n_segment, c, hw]; fold = c / fold_div; first fold channels come from t+1, next fold from t-1 */
void ref_temporal_shift(const float* x, int nt, int c, int hw, int n_segment, int fold_div, float* out) {
    const int fold = c / fold_div;
    memset(out, 0, (size_t)nt * c * hw * sizeof(float));
    for (int f = 0; f < nt; ++f) {
        const int t = f % n_segment;
        for (int ch = 0; ch < c; ++ch) {
            int src = f;
            if (ch < fold) { if (t == n_segment - 1) continue; src = f + 1; }
            else if (ch < 2 * fold) { if (t == 0) continue; src = f - 1; }
            memcpy(out + ((size_t)f * c + ch) * hw, x + ((size_t)src * c + ch) * hw, (size_t)hw * sizeof(float));
        }
    }
}
