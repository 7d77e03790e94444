// debug probe: the DDT integrand of bdrt_matrices.hip (blocking-planar, charge transfer) on the device against a long-double host
// evaluation, point by point of the quadrature.  Found the device library's sinh to be off by 2e-9 relative around |a| ~ 1e-8
// (with sinh(z.re) in ctanh_d: worst 2e-9; with the expm1 form: < 1e-15).  Build: hipcc --offload-arch=gfx950 tools/integrand_probe.hip
#include <hip/hip_runtime.h>
#include <complex>
#include <cstdio>
#include <cmath>
#define BDRT_KERNEL_DRT 0
#define BDRT_KERNEL_DDT_BLOCK_PLANAR 1
#define BDRT_KERNEL_DDT_BLOCK_SPHER 2
#define BDRT_BASIS_COLE_COLE 1
#define BDRT_BASIS_ZIC 2
namespace bdrt {
struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cdivi(cplx a, cplx b)
{
    if (fabs(b.re) >= fabs(b.im)) { const double r = b.im / b.re, d = b.re + b.im * r; return {(a.re + a.im * r) / d, (a.im - a.re * r) / d}; }
    const double r = b.re / b.im, d = b.re * r + b.im; return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ cplx csqrt_d(cplx z)
{
    const double m = hypot(z.re, z.im);
    if (m == 0.0) return {0.0, 0.0};
    if (z.re >= 0.0) { const double t = sqrt(0.5 * (m + z.re)); return {t, z.im / (2.0 * t)}; }
    const double t = sqrt(0.5 * (m - z.re)); return {fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
__device__ __forceinline__ cplx ctanh_d(cplx z)
{
    const double em = expm1(fabs(z.re));
    const double sh = copysign(0.5 * (em + em / (em + 1.0)), z.re);
    const double t = tan(z.im), beta = 1.0 + t * t, rho = sqrt(1.0 + sh * sh);
    const double den = 1.0 + beta * sh * sh;
    return {beta * rho * sh / den, t / den};
}
}
using namespace bdrt;
__global__ void probe(double w_n, double t_m, double eps, double k_ct, int n, double *out)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    const double y = -20.0 + i * (40.0 / 999);
    const double ey = exp(y);
    cplx arg = {t_m * ey * k_ct, t_m * ey * w_n};
    const cplx x = csqrt_d(arg);
    const cplx th = ctanh_d(x);
    const cplx one = {1.0, 0.0};
    cplx ZD = cdivi(one, cmul(th, x));
    out[6 * i] = ZD.re; out[6 * i + 1] = ZD.im; out[6 * i + 2] = x.re; out[6 * i + 3] = x.im; out[6 * i + 4] = th.re; out[6 * i + 5] = th.im;
}
int main()
{
    const double f = 4.70406202e-02, w = 2 * M_PI * f, tm = 1.04362667e-09, k = 0.41863750085857987;
    const int n = 1000;
    double *d; hipMalloc(&d, 6 * n * sizeof(double));
    probe<<<1, 1024>>>(w, tm, 0.19, k, n, d);
    static double h[6000]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); double worst = 0; int wi = 0;
    for (int i = 0; i < n; ++i) {
        const long double y = -20.0 + i * (40.0 / 999);
        std::complex<long double> arg = (long double)tm * expl(y) * std::complex<long double>(k, w);
        std::complex<long double> x = std::sqrt(arg), th = std::tanh(x), ZD = 1.0L / (x * th);
        const double r = (double)(std::abs(std::complex<long double>(h[6 * i], h[6 * i + 1]) - ZD) / std::abs(ZD));
        if (r > worst) { worst = r; wi = i; }
        if (i == 128) {
            std::complex<long double> xd(h[6 * i + 2], h[6 * i + 3]), thd(h[6 * i + 4], h[6 * i + 5]);
            printf("i 128: device (th - x)/x = (%.3Le, %.3Le); host long double (th - x)/x = (%.3Le, %.3Le); double ctanh: ", ((thd - xd) / xd).real(), ((thd - xd) / xd).imag(),
                   ((th - x) / x).real(), ((th - x) / x).imag());
            std::complex<double> x2((double)x.real(), (double)x.imag()), t2 = std::tanh(x2);
            printf("(%.3e, %.3e)\n", ((t2 - x2) / x2).real(), ((t2 - x2) / x2).imag());
        }
        if (i % 100 == 0 || r > 1e-13) printf("i %d y %.3f ZD ref (%.6Lg, %.6Lg) rel %.3e | x rel %.3e th rel %.3e |x| %.3Lg\n", i, (double)y, ZD.real(), ZD.imag(), r,
               (double)(std::abs(std::complex<long double>(h[6 * i + 2], h[6 * i + 3]) - x) / std::abs(x)),
               (double)(std::abs(std::complex<long double>(h[6 * i + 4], h[6 * i + 5]) - th) / std::abs(th)), std::abs(x));
    }
    printf("worst %.3e at i %d\n", worst, wi);
    return 0;
}
