// bdrt_matrices.hip -- A / L / M matrix construction on the GPU (include/bdrt.h section (1)).
//
// Replaces bayes_drt/matrices.py construct_A (:120-265), construct_L (:268-325), construct_M (:366-411).
// construct_A's definition is a 1000-point trapezoid on y = linspace(-20, 20, 1000) (matrices.py:236-238,
// :262-263): one wavefront per matrix entry, 64 lanes stride the quadrature grid (exp / complex tanh heavy),
// wave-shuffle reduction in a fixed order (deterministic).  Log-uniform grids take the Toeplitz path of the
// reference (first column + first row, matrices.py:213-242): nf + k entries instead of nf * k.
#include <cmath>

#include "bdrt_host.h"

namespace bdrt {

constexpr int NQUAD = 1000;

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cdivi(cplx a, cplx b)
{
    // Smith's algorithm
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, d = b.re + b.im * r;
        return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
    }
    const double r = b.re / b.im, d = b.re * r + b.im;
    return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ cplx csqrt_d(cplx z)
{
    const double m = hypot(z.re, z.im);
    if (m == 0.0) return {0.0, 0.0};
    if (z.re >= 0.0) {
        const double t = sqrt(0.5 * (m + z.re));
        return {t, z.im / (2.0 * t)};
    }
    const double t = sqrt(0.5 * (m - z.re));
    return {fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
// tanh(a + ib) = (sinh 2a + i sin 2b) / (cosh 2a + cos 2b); |a| large -> +-1 without overflow
__device__ __forceinline__ cplx ctanh_d(cplx z)
{
    if (fabs(z.re) > 22.0) {
        const double e = exp(-2.0 * fabs(z.re));
        return {copysign(1.0, z.re), 4.0 * sin(z.im) * cos(z.im) * e};
    }
    // Kahan's algorithm (what C's ctanh implements): with t = tan b, beta = 1 + t^2, s = sinh a, rho = sqrt(1 + s^2):
    // tanh(a + ib) = (beta rho s + i t) / (1 + beta s^2).  sinh from expm1: the device library's sinh(a) is off by up to 2e-9
    // relative for |a| ~ 1e-8 .. 1e-7 (tools/integrand_probe.hip: (tanh x - x) / x = 1.9e-9 at |x| = 1.4e-8, where it is 6e-17),
    // which a randomised sweep caught as 6e-11 in a DDT matrix entry dominated by that stretch of the quadrature
    // (profiles/r03/fuzz_parity.txt, case 3167)
    const double em = expm1(fabs(z.re));
    const double sh = copysign(0.5 * (em + em / (em + 1.0)), z.re);
    const double t = tan(z.im), beta = 1.0 + t * t, rho = sqrt(1.0 + sh * sh);
    if (isinf(t)) return {rho / sh, 1.0 / t};
    const double den = 1.0 + beta * sh * sh;
    return {beta * rho * sh / den, t / den};
}

// get_basis_func (matrices.py:8-24): gaussian (:12-13), Cole-Cole (:15-17), Zic (:19-21; epsilon unused)
__device__ __forceinline__ double basis_phi(double y, double eps, int basis)
{
    if (basis == BDRT_BASIS_COLE_COLE)
        return (1.0 / (2.0 * M_PI)) * sin((1.0 - eps) * M_PI) / (cosh(eps * y) - cos((1.0 - eps) * M_PI));
    if (basis == BDRT_BASIS_ZIC) return 2.0 * exp(y) / (1.0 + exp(2.0 * y));
    return exp(-(eps * y) * (eps * y));
}

// integrand of get_A_func (matrices.py:27-117)
__device__ __forceinline__ double integrand(double y, double w_n, double t_m, double eps, int kernel, int part,
                                            int dist_series, int use_ct, double k_ct, int basis)
{
    const double phi = basis_phi(y, eps, basis);
    if (kernel == BDRT_KERNEL_DRT) {
        const double den = 1.0 + exp(2.0 * (y + log(w_n * t_m)));
        if (part == 0) return phi / den;                          // :48-49
        return -phi * exp(y) * w_n * t_m / den;                   // :51-52
    }
    const double ey = exp(y);
    cplx arg = use_ct ? cplx{t_m * ey * k_ct, t_m * ey * w_n} : cplx{0.0, w_n * t_m * ey};
    const cplx x = csqrt_d(arg);
    const cplx th = ctanh_d(x);
    cplx ZD;
    const cplx one = {1.0, 0.0};
    if (kernel == BDRT_KERNEL_DDT_BLOCK_PLANAR) ZD = cdivi(one, cmul(th, x));          // :62-70
    else if (kernel == BDRT_KERNEL_DDT_BLOCK_SPHER) {                                   // :74-80: tanh x / (x - tanh x)
        const double u2 = x.re * x.re + x.im * x.im;
        if (u2 < 1e-3) {
            // x - tanh x = x^3/3 - ...: below |x| ~ 1e-5 the difference of the two rounded numbers is zero or noise (the
            // reference's own values are noise there).  Laurent series 3/x^2 + 1/5 - x^2/175 (next term 2 x^4 / 7875)
            const cplx xx = cmul(x, x);
            const cplx inv = cdivi(cplx{3.0, 0.0}, xx);
            ZD = {inv.re + 0.2 - xx.re * (1.0 / 175.0), inv.im - xx.im * (1.0 / 175.0)};
        } else {
            ZD = cdivi(th, cplx{x.re - th.re, x.im - th.im});
        }
    }
    else ZD = cdivi(th, x);                                                            // :86-92
    const cplx val = dist_series ? ZD : cdivi(one, ZD);                                 // :97-110
    return phi * (part == 0 ? val.re : val.im);
}

__device__ __forceinline__ double quad_y(int i)
{
    return (i == NQUAD - 1) ? 20.0 : -20.0 + i * (40.0 / (NQUAD - 1));   // np.linspace(-20, 20, 1000)
}

// one wave per entry.  toeplitz: entry e < nf -> (w_e, t_0), else (w_0, t_{e-nf}).  general: (w_{e/k}, t_{e%k})
__global__ __launch_bounds__(256) void build_A_entries(const double *freq, int nf, const double *tau, int k, double eps,
                                                       int kernel, int part, int dist_series, int use_ct, double k_ct,
                                                       int toeplitz, int nentries, int basis, double *vals)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nentries) return;
    int n, m;
    if (toeplitz) { n = wave < nf ? wave : 0; m = wave < nf ? 0 : wave - nf; }
    else { n = wave / k; m = wave % k; }
    const double w_n = freq[n] * 2.0 * M_PI, t_m = tau[m];
    // trapezoid: sum_i (y_{i+1} - y_i) (f_i + f_{i+1}) / 2  ==  sum_i f_i (y_{i+1} - y_{i-1}) / 2 with one-sided ends
    double s = 0.0;
    for (int i = lane; i < NQUAD; i += 64) {
        const double y = quad_y(i);
        const double lo = i > 0 ? quad_y(i - 1) : y, hi = i < NQUAD - 1 ? quad_y(i + 1) : y;
        s += integrand(y, w_n, t_m, eps, kernel, part, dist_series, use_ct, k_ct, basis) * (0.5 * (hi - lo));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) vals[wave] = s;
}

__global__ void toeplitz_expand(const double *vals, int nf, int k, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf * k) return;
    const int n = i / k, m = i % k;
    out[i] = n >= m ? vals[n - m] : vals[nf + (m - n)];      // scipy.linalg.toeplitz(c, r)
}

// L[n,m] = sum_j coef[j] d^j/dy^j exp(-(eps y)^2), y = ln(1/(w_n t_m))   (matrices.py:268-325)
// freq == nullptr: the collocated case w_n = 2 pi (1/(2 pi t_n)) Inverter uses (nf = k); else any frequencies, [nf x k].
// Zic basis: the function itself, order 0 only (matrices.py:316-318).
__global__ void build_L_kernel(const double *freq, int nf, const double *tau, int k, double eps, double c0, double c1, double c2,
                               double c3, int basis, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf * k) return;
    const int n = i / k, m = i % k;
    const double f_n = freq ? freq[n] : 1.0 / (2.0 * M_PI * tau[n]);
    const double w_n = 2.0 * M_PI * f_n;
    const double y = log(1.0 / (w_n * tau[m]));
    if (basis == BDRT_BASIS_ZIC) { out[i] = c0 * basis_phi(y, eps, basis); return; }
    const double g = exp(-(eps * y) * (eps * y));
    const double e2 = eps * eps;
    double v = 0.0;
    if (c0 != 0.0) v += c0 * g;
    if (c1 != 0.0) v += c1 * (-2.0 * e2 * y * g);
    if (c2 != 0.0) v += c2 * ((-2.0 * e2 + 4.0 * e2 * e2 * y * y) * g);
    if (c3 != 0.0) v += c3 * ((12.0 * e2 * e2 * y - 8.0 * e2 * e2 * e2 * y * y * y) * g);
    out[i] = v;
}

// closed-form penalty matrices (matrices.py:328-411)
__global__ void build_M_kernel(const double *tau, int k, double eps, double c0, double c1, double c2, int toeplitz,
                               double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k * k) return;
    int n = i / k, m = i % k;
    if (toeplitz) { n = abs(n - m); m = 0; }                  // symmetric toeplitz(c), c[n] = func(w_n, t_0)  (:396-405)
    const double w_n = (1.0 / (2.0 * M_PI * tau[n])) * 2.0 * M_PI;
    const double w_m = (1.0 / (2.0 * M_PI * tau[m])) * 2.0 * M_PI;
    const double t_m = 1.0 / w_m;
    const double a = eps * log(1.0 / (w_n * t_m));
    const double g = exp(-(a * a / 2.0));
    const double rt = sqrt(M_PI / 2.0);
    double v = 0.0;
    if (c0 != 0.0) v += c0 * (rt / eps * g);
    if (c1 != 0.0) v += c1 * (-rt * eps * (-1.0 + a * a) * g);
    if (c2 != 0.0) v += c2 * (rt * eps * eps * eps * (3.0 - 6.0 * a * a + a * a * a * a) * g);
    out[i] = v;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { BDRT_HIP(hipMalloc(&p, bytes ? bytes : 8)); return 0; }
    template <class T> T *as() { return (T *)p; }
};

}  // namespace bdrt

using namespace bdrt;

extern "C" {

int bdrt_build_A(const double *freq, int nf, const double *tau, int k, double eps, int kernel_id, int part,
                 int dist_series, int use_ct, double k_ct, int toeplitz, double *out)
{
    return bdrt_build_A_basis(freq, nf, tau, k, eps, kernel_id, part, dist_series, use_ct, k_ct, toeplitz, BDRT_BASIS_GAUSSIAN, out);
}

int bdrt_build_A_basis(const double *freq, int nf, const double *tau, int k, double eps, int kernel_id, int part,
                       int dist_series, int use_ct, double k_ct, int toeplitz, int basis_id, double *out)
{
    if (!freq || !tau || !out || nf <= 0 || k <= 0 || kernel_id < 0 || kernel_id > 3 || (part != 0 && part != 1) ||
        basis_id < BDRT_BASIS_GAUSSIAN || basis_id > BDRT_BASIS_ZIC) {
        set_error("bdrt_build_A: bad arguments");
        return -1;
    }
    if (basis_id == BDRT_BASIS_COLE_COLE && !(eps > 0.0 && eps < 1.0)) {
        set_error("bdrt_build_A: the Cole-Cole basis needs 0 < epsilon < 1");
        return -1;
    }
    bind_process_device();
    DevBuf dF, dT, dV, dO;
    int rc;
    if ((rc = dF.alloc(nf * sizeof(double))) || (rc = dT.alloc(k * sizeof(double)))) return rc;
    const int nent = toeplitz ? nf + k : nf * k;
    if ((rc = dV.alloc((size_t)nent * sizeof(double))) || (rc = dO.alloc((size_t)nf * k * sizeof(double)))) return rc;
    BDRT_HIP(hipMemcpy(dF.p, freq, nf * sizeof(double), hipMemcpyHostToDevice));
    BDRT_HIP(hipMemcpy(dT.p, tau, k * sizeof(double), hipMemcpyHostToDevice));
    const int wpb = 4;   // waves per block
    hipLaunchKernelGGL(build_A_entries, dim3((nent + wpb - 1) / wpb), dim3(64 * wpb), 0, 0, dF.as<double>(), nf,
                       dT.as<double>(), k, eps, kernel_id, part, dist_series, use_ct, k_ct, toeplitz, nent, basis_id,
                       toeplitz ? dV.as<double>() : dO.as<double>());
    BDRT_HIP(hipGetLastError());
    if (toeplitz) {
        double c0, r0;
        BDRT_HIP(hipMemcpy(&c0, dV.as<double>(), sizeof(double), hipMemcpyDeviceToHost));
        BDRT_HIP(hipMemcpy(&r0, dV.as<double>() + nf, sizeof(double), hipMemcpyDeviceToHost));
        if (!(c0 == r0)) {                       // matrices.py:239-241
            set_error("First entries of first row and column are not equal (%.17g vs %.17g)", r0, c0);
            return -2;
        }
        hipLaunchKernelGGL(toeplitz_expand, dim3((nf * k + 255) / 256), dim3(256), 0, 0, dV.as<double>(), nf, k,
                           dO.as<double>());
        BDRT_HIP(hipGetLastError());
    }
    BDRT_HIP(hipMemcpy(out, dO.p, (size_t)nf * k * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int bdrt_build_L(const double *tau, int k, double eps, const double *coef4, double *out)
{
    return bdrt_build_L_rect(nullptr, k, tau, k, eps, coef4, BDRT_BASIS_GAUSSIAN, out);
}

int bdrt_build_L_rect(const double *freq, int nf, const double *tau, int k, double eps, const double *coef4, int basis_id,
                      double *out)
{
    if (!tau || !coef4 || !out || k <= 0 || nf <= 0 || (!freq && nf != k) ||
        (basis_id != BDRT_BASIS_GAUSSIAN && basis_id != BDRT_BASIS_ZIC)) {
        set_error("bdrt_build_L: bad arguments");
        return -1;
    }
    if (basis_id == BDRT_BASIS_ZIC && (coef4[1] != 0.0 || coef4[2] != 0.0 || coef4[3] != 0.0)) {
        set_error("bdrt_build_L: the Zic basis has order 0 only (matrices.py:316-318)");
        return -1;
    }
    bind_process_device();
    DevBuf dF, dT, dO;
    int rc;
    if ((rc = dT.alloc(k * sizeof(double))) || (rc = dO.alloc((size_t)nf * k * sizeof(double)))) return rc;
    if (freq) {
        if ((rc = dF.alloc(nf * sizeof(double)))) return rc;
        BDRT_HIP(hipMemcpy(dF.p, freq, nf * sizeof(double), hipMemcpyHostToDevice));
    }
    BDRT_HIP(hipMemcpy(dT.p, tau, k * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(build_L_kernel, dim3((nf * k + 255) / 256), dim3(256), 0, 0, freq ? dF.as<double>() : (const double *)nullptr,
                       nf, dT.as<double>(), k, eps, coef4[0], coef4[1], coef4[2], coef4[3], basis_id, dO.as<double>());
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipMemcpy(out, dO.p, (size_t)nf * k * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int bdrt_build_M(const double *tau, int k, double eps, const double *coef3, int toeplitz, double *out)
{
    if (!tau || !coef3 || !out || k <= 0) { set_error("bdrt_build_M: bad arguments"); return -1; }
    bind_process_device();
    DevBuf dT, dO;
    int rc;
    if ((rc = dT.alloc(k * sizeof(double))) || (rc = dO.alloc((size_t)k * k * sizeof(double)))) return rc;
    BDRT_HIP(hipMemcpy(dT.p, tau, k * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(build_M_kernel, dim3((k * k + 255) / 256), dim3(256), 0, 0, dT.as<double>(), k, eps, coef3[0],
                       coef3[1], coef3[2], toeplitz, dO.as<double>());
    BDRT_HIP(hipGetLastError());
    BDRT_HIP(hipMemcpy(out, dO.p, (size_t)k * k * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
