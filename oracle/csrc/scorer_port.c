/* oracle/csrc/scorer_port.c — plain-C restatement of the reference coverage-gain scorer.
 *
 * TEST INFRASTRUCTURE ONLY: used as the checker in tests/ and as bench.py's timed `cpu_baseline`
 * (kind "port").  Never linked into libmacarons_hip.so.
 *
 * Follows, per (camera, point) pair and in fp32 like the reference:
 *   macarons/networks/SconeVis.py:230-233   rays, spherical coords, theta = pi/2 - elev
 *   macarons/utility/CustomGeometry.py:27-45 asin / acos with clamps, sign from x
 *   macarons/utility/spherical_harmonics.py:67-140  lpmv recursion + cos(m phi) / sin(|m| phi), norms
 *   macarons/networks/SconeVis.py:241-250   dot with the 64 coefficients, sigmoid|relu, mean over points
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#include <stdlib.h>
#endif

#define LMAX 8
static float g_norm[LMAX][LMAX];
static float g_pmm[LMAX];
static int g_init = 0;

static double semifact(int x) { double r = 1; for (int v = x; v > 1; v -= 2) r *= v; return r; }
static double poch(int x, int k) { double r = x; for (int v = x + 1; v < x + k; ++v) r *= v; return r; }

static void init_tables(void) {
    if (g_init) return;
    for (int l = 0; l < LMAX; ++l)
        for (int m = 0; m <= l; ++m) {
            double n = sqrt((2 * l + 1) / (4 * M_PI));
            if (m) n *= sqrt(2.0 / poch(l - m + 1, 2 * m));
            g_norm[l][m] = (float)n;
        }
    for (int m = 0; m < LMAX; ++m) g_pmm[m] = (float)((m & 1 ? -1.0 : 1.0) * semifact(2 * m - 1));
    g_init = 1;
}

static inline float pair_value(float dx, float dy, float dz, const float* h, int use_sigmoid) {
    const float PI = 3.14159265358979323846f;
    float r = sqrtf(dx * dx + dy * dy + dz * dz);
    float yr = dy / r;
    float elev = asinf(yr);
    if (yr <= -1.f) elev = -PI / 2;
    if (yr >= 1.f) elev = PI / 2;
    float q = dz / (r * cosf(elev));
    float azim = acosf(q);
    if (q <= -1.f) azim = PI;
    if (q >= 1.f) azim = 0.f;
    if (dx < 0.f) azim = -azim;
    float theta = -elev + PI / 2.f;
    float x = cosf(theta);
    float om = 1.f - x * x;
    float P[LMAX][LMAX];
    for (int m = 0; m < LMAX; ++m) {
        P[m][m] = m == 0 ? 1.f : g_pmm[m] * powf(om, 0.5f * (float)m);
        for (int l = m + 1; l < LMAX; ++l) {
            float y = ((float)(2 * l - 1) / (float)(l - m)) * x * P[l - 1][m];
            if (l - m > 1) y -= ((float)(l + m - 1) / (float)(l - m)) * P[l - 2][m];
            P[l][m] = y;
        }
    }
    float z = 0.f;
    for (int l = 0; l < LMAX; ++l)
        for (int m = -l; m <= l; ++m) {
            int a = m < 0 ? -m : m;
            float Y;
            if (m == 0) Y = g_norm[l][0] * P[l][0];
            else Y = (m > 0 ? cosf((float)m * azim) : sinf((float)a * azim)) * P[l][a] * g_norm[l][a];
            z += Y * h[l * l + l + m];
        }
    return use_sigmoid ? 1.f / (1.f + expf(-z)) : (z > 0.f ? z : 0.f);
}

/* Thread count of the next calls (0 = leave OpenMP's default).  The benchmark asks for every host core. */
static int g_threads = 0;
void scorer_port_set_threads(int n) { g_threads = n; }

/* gains[B,C];  returns the number of threads the parallel region actually ran with.
 * Work items = (cloud-camera pair, chunk of CHUNK points): 200 cameras alone would leave most of a 256-core host idle.  Every
 * chunk keeps the blocked fp32 sums (1024 points, like torch.sum) in a double; a pair's chunks are added in order. */
#define CHUNK 8192
int scorer_port_coverage_gain(const float* pts, int pts_dim, const float* harm, const float* cams, float* gains,
                              int64_t B, int64_t N, int64_t C, int use_sigmoid) {
    init_tables();
    int nthreads = 1;
    const int64_t BC = B * C, NCH = (N + CHUNK - 1) / CHUNK;
    double* part = (double*)malloc((size_t)(BC * NCH) * sizeof(double));
    if (!part) return -1;
#ifdef _OPENMP
    if (g_threads > 0) omp_set_num_threads(g_threads);
#endif
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nthreads = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 1)
        for (int64_t w = 0; w < BC * NCH; ++w) {
            const int64_t bc = w / NCH, ch = w - bc * NCH, b = bc / C;
            const float* cam = cams + bc * 3;
            const int64_t n0 = ch * CHUNK, n1 = n0 + CHUNK < N ? n0 + CHUNK : N;
            double acc = 0.0;
            float accf = 0.f;
            for (int64_t n = n0; n < n1; ++n) {
                const float* p = pts + (b * N + n) * pts_dim;
                accf += pair_value(cam[0] - p[0], cam[1] - p[1], cam[2] - p[2], harm + (b * N + n) * 64, use_sigmoid);
                if ((n & 1023) == 1023) { acc += accf; accf = 0.f; }   /* blocked fp32 sum (torch.sum is blocked too) */
            }
            part[w] = acc + accf;
        }
    }
    for (int64_t bc = 0; bc < BC; ++bc) {
        double acc = 0.0;
        for (int64_t ch = 0; ch < NCH; ++ch) acc += part[bc * NCH + ch];
        gains[bc] = (float)(acc / (double)N);
    }
    free(part);
    return nthreads;
}

/* vis[B,C,N] */
int scorer_port_visibilities(const float* pts, int pts_dim, const float* harm, const float* cams, float* vis,
                             int64_t B, int64_t N, int64_t C, int use_sigmoid) {
    init_tables();
    const int64_t BC = B * C;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t bc = 0; bc < BC; ++bc) {
        const int64_t b = bc / C;
        const float* cam = cams + bc * 3;
        for (int64_t n = 0; n < N; ++n) {
            const float* p = pts + (b * N + n) * pts_dim;
            vis[bc * N + n] = pair_value(cam[0] - p[0], cam[1] - p[1], cam[2] - p[2], harm + (b * N + n) * 64, use_sigmoid);
        }
    }
    return 0;
}
