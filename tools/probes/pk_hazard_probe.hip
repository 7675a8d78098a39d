// Probe: are back-to-back dependent v_pk_mul_f32 -> v_pk_fma_f32 (no s_nop between) safe on gfx950?
// hipcc inserts "s_nop 0" between dependent packed-f32 ops; this checks the hardware result without it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float2 *a, const float2 *b, float2 *out, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    v2f x = {a[i].x, a[i].y}, w = {b[i].x, b[i].y};
    for (int it = 0; it < iters; it++) {
        v2f t, r;
        asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
                     "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                     : "=&v"(t), "=&v"(r) : "v"(x), "v"(w));
        // dependent chain straight into the next pair
        asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 %0, %0, %2, %1 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
                     : "=&v"(x) : "v"(r), "v"(w));
    }
    out[i] = make_float2(x.x, x.y);
}
static void ref(float2 &x, float2 w, int iters)
{
    for (int it = 0; it < iters; it++) {
        float tx = x.x * w.x, ty = x.x * w.y;
        float rx = fmaf(x.y, -w.y, tx), ry = fmaf(x.y, w.x, ty);
        float ux = fmaf(rx, w.x, rx), uy = fmaf(ry, w.x, ry);        // src1 op_sel_hi 0 -> w.x for both halves
        ux = fmaf(ux, w.x, rx); uy = fmaf(uy, w.x, ry);
        x.x = ux + ry; x.y = uy - rx;                                // + (r.y, -r.x)
    }
}
int main()
{
    const int n = 256 * 1024, iters = 7;
    std::vector<float2> a(n), b(n), o(n);
    for (int i = 0; i < n; i++) { a[i] = make_float2(sinf(i * 0.37f), cosf(i * 0.11f)); b[i] = make_float2(0.6f * cosf(i * 0.7f), 0.6f * sinf(i * 0.7f)); }
    float2 *da, *db, *dout;
    (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dout, n * 8);
    (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dout, iters);
    (void)hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0; double maxd = 0;
    for (int i = 0; i < n; i++) {
        float2 x = a[i]; ref(x, b[i], iters);
        double d = fmax(fabs((double)x.x - o[i].x), fabs((double)x.y - o[i].y));
        if (d > maxd) maxd = d;
        if (x.x != o[i].x || x.y != o[i].y) bad++;
    }
    printf("pk hazard probe: %d of %d lanes differ bitwise from the scalar fmaf reference, max abs diff %.3g\n", bad, n, maxd);
    return bad != 0;
}
