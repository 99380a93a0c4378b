// Complex (c128) side of libkrylov_hip.
//
// A complex N-vector is held as a REAL kh_vec of length 2N (interleaved re, im): every operation
// that is real-linear - copy, zero, 2-norm, scaling by a real, axpby with real coefficients, the
// normalise-and-store of the Arnoldi step - is the existing real kernel on that view, untouched.
// This file adds only what is genuinely complex: conj(V)^T w panel products, updates with complex
// coefficients, complex CSR / dense / diagonal operators, and the Arnoldi step built from them.
// Arithmetic follows NumPy's complex formulas ((ar*br - ai*bi), (ar*bi + ai*br), separate
// roundings; -ffp-contract=off), CSR rows are summed left to right like scipy's csr_matvec.
// Correctness first: per-column launches (no register-resident chain yet).
#pragma once
// included at the end of krylov_hip.hip (one translation unit: the kernels of kernels.h are shared)

namespace kh {

constexpr int ZMAXC = 8;   // complex columns per panel launch (2 partial-sum slots each)
constexpr int ZMAXD = 512; // widest complex projector (kh_zproj_create)

struct ZColPtrs {
    const double2* c[ZMAXC];
};

// part_out[(2c) * pstride + wg] = Re partial, [(2c+1) * pstride + wg] = Im partial of <v_c, w>
template <int C>
__global__ __launch_bounds__(BS) void k_zmultidot(int64_t n, ZColPtrs cols, const double2* __restrict__ w,
                                                  double* __restrict__ part_out, int pstride) {
    __shared__ double sm[8];
    const int64_t stride = (int64_t)gridDim.x * BS;
    double ar[C], ai[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ar[c] = ai[c] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += stride) {
        const double2 wv = w[i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double2 v = ld_nt2(cols.c[c] + i);    // column: used once per launch
            ar[c] = fma(v.x, wv.x, ar[c]);     // conj(v) * w
            ar[c] = fma(v.y, wv.y, ar[c]);
            ai[c] = fma(v.x, wv.y, ai[c]);
            ai[c] = fma(-v.y, wv.x, ai[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double r = block_sum(ar[c], sm);
        if (threadIdx.x == 0) part_out[(int64_t)(2 * c) * pstride + blockIdx.x] = r;
        __syncthreads();
        r = block_sum(ai[c], sm);
        if (threadIdx.x == 0) part_out[(int64_t)(2 * c + 1) * pstride + blockIdx.x] = r;
        __syncthreads();
    }
}

// w = beta*w - sign * sum_c coef[c] * v_c (complex coefficients, left to right); optional <w,w> partials
template <int C, bool NRM, int BETA>
__global__ __launch_bounds__(BS) void k_zmultiaxpy(int64_t n, ZColPtrs cols, const double2* __restrict__ coef,
                                                   double sign, double beta, double2* __restrict__ w,
                                                   double* __restrict__ part_out) {
    __shared__ double sm[8];
    double2 h[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        h[c].x = sign * coef[c].x;
        h[c].y = sign * coef[c].y;
    }
    const int64_t stride = (int64_t)gridDim.x * BS;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += stride) {
        double2 wv;
        if (BETA == 0) {
            wv.x = 0.0;
            wv.y = 0.0;
        } else {
            wv = w[i];
            if (BETA == 2) {
                wv.x = beta * wv.x;
                wv.y = beta * wv.y;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double2 v = ld_nt2(cols.c[c] + i);    // column: used once per launch
            const double tr = h[c].x * v.x - h[c].y * v.y;
            const double ti = h[c].x * v.y + h[c].y * v.x;
            wv.x = wv.x - tr;
            wv.y = wv.y - ti;
        }
        st_nt2(w + i, wv);
        if (NRM) {
            acc = fma(wv.x, wv.x, acc);
            acc = fma(wv.y, wv.y, acc);
        }
    }
    if (NRM) {
        const double r = block_sum(acc, sm);
        if (threadIdx.x == 0) part_out[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(BS) void k_zwaxpby(int64_t n, double2* z, double2 alpha, const double2* x,
                                                double2 beta, const double2* y) {
    const int64_t stride = (int64_t)gridDim.x * BS;
    const bool b0 = (beta.x == 0.0 && beta.y == 0.0);
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += stride) {
        const double2 xv = x[i];
        double2 r;
        r.x = alpha.x * xv.x - alpha.y * xv.y;
        r.y = alpha.x * xv.y + alpha.y * xv.x;
        if (!b0) {
            const double2 yv = y[i];
            r.x = r.x + (beta.x * yv.x - beta.y * yv.y);
            r.y = r.y + (beta.x * yv.y + beta.y * yv.x);
        }
        st_nt2(z + i, r);
    }
}

// Complex twin of k_minres_update (linsys.py:844-846), one pass:
//   z = ((v - r0 w0) - r1 w1) / r2 ;  w0 <- z ;  yk += y0 z
// NumPy's complex arithmetic: products as (ac - bd, ad + bc), the quotient by Smith's formula (ratio and scale once).
__global__ __launch_bounds__(BS) void k_zminres_update(int64_t n, const double2* __restrict__ v,
                                                       double2* __restrict__ w0, const double2* __restrict__ w1,
                                                       double2 r0, double2 r1, double2 r2, double2 y0,
                                                       double2* __restrict__ yk) {
    const bool big_re = fabs(r2.x) >= fabs(r2.y);
    const double rat = big_re ? r2.y / r2.x : r2.x / r2.y;
    const double scl = 1.0 / (big_re ? (r2.x + r2.y * rat) : (r2.y + r2.x * rat));
    const int64_t stride = (int64_t)gridDim.x * BS;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += stride) {
        const double2 a = w0[i], b = w1[i];
        double2 t = v[i];
        t.x = t.x - (r0.x * a.x - r0.y * a.y);
        t.y = t.y - (r0.x * a.y + r0.y * a.x);
        t.x = t.x - (r1.x * b.x - r1.y * b.y);
        t.y = t.y - (r1.x * b.y + r1.y * b.x);
        double2 z;
        if (big_re) {
            z.x = (t.x + t.y * rat) * scl;
            z.y = (t.y - t.x * rat) * scl;
        } else {
            z.x = (t.x * rat + t.y) * scl;
            z.y = (t.y * rat - t.x) * scl;
        }
        st_nt2(w0 + i, z);
        double2 o = yk[i];
        o.x = o.x + (y0.x * z.x - y0.y * z.y);
        o.y = o.y + (y0.x * z.y + y0.y * z.x);
        st_nt2(yk + i, o);
    }
}

// Complex CG: t[2], t[3] = <p, Ap>; t[0] <- the real number d with rho / d = Re(rho / <p, Ap>), the step length the
// reference takes (linsys.py:640-648: complex quotient, then the real part) - Smith's formula with a real numerator.
__global__ void k_zcg_den(double* __restrict__ t) {
    const double re = t[2], im = t[3];
    if (fabs(re) >= fabs(im)) {
        t[0] = re + im * (im / re);
    } else {
        const double ratio = re / im;
        t[0] = (re * ratio + im) / ratio;
    }
}

__global__ __launch_bounds__(BS) void k_zdiag_apply(int64_t n, const double2* __restrict__ d,
                                                    const double2* __restrict__ x, double2* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * BS;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += stride) {
        const double2 a = d[i], b = x[i];
        double2 r;
        r.x = a.x * b.x - a.y * b.y;
        r.y = a.x * b.y + a.y * b.x;
        st_nt2(y + i, r);
    }
}

// diagonal-major copy of a banded complex CSR operator: zdia[d * ld + i] = A[i, i + off[d]] ((0, 0) where the row has no entry)
static __global__ __launch_bounds__(BS) void k_zdia_fill(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                        const double2* __restrict__ data, int64_t n_rows, DiaOffs o,
                                                        double2* __restrict__ zdia, int64_t ld) {
    const int64_t r = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (r >= n_rows) return;
    for (int p = indptr[r]; p < indptr[r + 1]; ++p) {
        const int off = indices[p] - (int)r;
        int d = 0;
        while (d < o.nd - 1 && o.off[d] != off) ++d;
        zdia[(int64_t)d * ld + r] = data[p];
    }
}

// complex CSR-stream SpMV: products parked in LDS as double2, rows summed left to right
template <int ITEMS>
__global__ __launch_bounds__(BS) void k_zspmv_stream(const int32_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices,
                                                     const double2* __restrict__ data,
                                                     const int32_t* __restrict__ rowblk, int nblk, int tile,
                                                     const double2* __restrict__ x, double2* __restrict__ y,
                                                     int64_t nloc, const double2* __restrict__ ghost) {
    // columns >= nloc of a block-row shard are ghost entries (rows of the neighbouring slabs, filled by the halo
    // exchange): the same layout as the real kernels
    extern __shared__ __attribute__((aligned(16))) double2 zprod[];
    __shared__ double sm[8];
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int r0 = rowblk[bid], r1 = rowblk[bid + 1];
    const int nz0 = indptr[r0], nz1 = indptr[r1];
    const int cnt = nz1 - nz0;
    if (cnt <= tile) {
        if (cnt > 0) {
            int c[ITEMS];
            double2 a[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int t = threadIdx.x + i * BS;
                const int tc = t < cnt ? t : cnt - 1;
                c[i] = __builtin_nontemporal_load(indices + nz0 + tc);   // read-once streams
                a[i] = ld_nt2(data + nz0 + tc);
            }
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const double2 xv = (c[i] < nloc) ? x[c[i]] : ghost[c[i] - nloc];
                double2 p;
                p.x = a[i].x * xv.x - a[i].y * xv.y;
                p.y = a[i].x * xv.y + a[i].y * xv.x;
                a[i] = p;
            }
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int t = threadIdx.x + i * BS;
                if (t < cnt) zprod[t] = a[i];
            }
        }
        __syncthreads();
        for (int r = r0 + threadIdx.x; r < r1; r += BS) {
            const int p0 = indptr[r] - nz0, p1 = indptr[r + 1] - nz0;
            double2 s;
            s.x = 0.0;
            s.y = 0.0;
            for (int p = p0; p < p1; ++p) {
                s.x += zprod[p].x;
                s.y += zprod[p].y;
            }
            st_nt2(y + r, s);
        }
    } else {  // one long row: tree reduction
        double sr = 0.0, si = 0.0;
        for (int t = threadIdx.x; t < cnt; t += BS) {
            const double2 a = data[nz0 + t];
            const int cc = indices[nz0 + t];
            const double2 xv = (cc < nloc) ? x[cc] : ghost[cc - nloc];
            sr += a.x * xv.x - a.y * xv.y;
            si += a.x * xv.y + a.y * xv.x;
        }
        sr = block_sum(sr, sm);
        __syncthreads();
        si = block_sum(si, sm);
        if (threadIdx.x == 0) {
            double2 s;
            s.x = sr;
            s.y = si;
            y[r0] = s;
        }
    }
}

// Stand-alone banded complex SpMV (round 6): y = A x from the diagonal-major copy of (re, im) pairs - 16 B per slot and no index
// stream instead of 20 B per entry, the x gather nd coalesced streams.  One lane per complex row, the diagonals in ascending
// offset = the storage order of a CSR row with sorted columns, NumPy's product formula, sums from (0, 0), empty slots skipped:
// the bits of k_zspmv_stream and of chain_apply_banded_z (the operator in the chain kernels' prologue).
template <int ND>
__global__ __launch_bounds__(BS) void k_zspmv_dia(DiaOffs o, const double2* __restrict__ zdia, int64_t ld, int64_t n_rows,
                                                  const double2* __restrict__ x, double2* __restrict__ y) {
    const int nd = ND > 0 ? ND : o.nd;
    const int64_t last = n_rows - 1;
    const int64_t stride = (int64_t)gridDim.x * BS;
    for (int64_t row = (int64_t)blockIdx.x * BS + threadIdx.x; row < n_rows; row += stride) {
        double sx = 0.0, sy = 0.0;
        if constexpr (ND > 0) {
            double2 av[ND > 0 ? ND : 1], xv[ND > 0 ? ND : 1];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                av[d] = ld_nt2(zdia + (int64_t)d * ld + row);
                int64_t c = row + o.off[d];
                c = c < 0 ? 0 : (c > last ? last : c);
                xv[d] = x[c];
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double px = av[d].x * xv[d].x - av[d].y * xv[d].y;
                const double py = av[d].x * xv[d].y + av[d].y * xv[d].x;
                const bool on = (av[d].x != 0.0) || (av[d].y != 0.0);
                sx = on ? sx + px : sx;
                sy = on ? sy + py : sy;
            }
        } else {
            for (int d = 0; d < nd; ++d) {
                const double2 a = ld_nt2(zdia + (int64_t)d * ld + row);
                if ((a.x != 0.0) || (a.y != 0.0)) {
                    int64_t c = row + o.off[d];
                    c = c < 0 ? 0 : (c > last ? last : c);
                    const double2 xv = x[c];
                    sx = sx + (a.x * xv.x - a.y * xv.y);
                    sy = sy + (a.x * xv.y + a.y * xv.x);
                }
            }
        }
        double2 s;
        s.x = sx;
        s.y = sy;
        st_nt2(y + row, s);
    }
}

__global__ __launch_bounds__(BS) void k_zgemv_dense(int64_t n_rows, int64_t n_cols, const double2* __restrict__ a,
                                                    int64_t lda, const double2* __restrict__ x,
                                                    double2* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const double2* __restrict__ ar = a + row * lda;
    double sr = 0.0, si = 0.0;
    for (int64_t i = lane; i < n_cols; i += 64) {
        const double2 av = ar[i], xv = x[i];
        sr = fma(av.x, xv.x, sr);
        sr = fma(-av.y, xv.y, sr);
        si = fma(av.x, xv.y, si);
        si = fma(av.y, xv.x, si);
    }
    sr = wave_sum(sr);
    si = wave_sum(si);
    if (lane == 0) {
        double2 s;
        s.x = sr;
        s.y = si;
        y[row] = s;
    }
}

// z = x + 0i (widen a real column to complex)
__global__ __launch_bounds__(BS) void k_zfrom_real(int64_t n, const double* __restrict__ x, double2* __restrict__ z) {
    const int64_t stride = (int64_t)gridDim.x * BS;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += stride) {
        double2 r;
        r.x = x[i];
        r.y = 0.0;
        st_nt2(z + i, r);
    }
}

// hdev[2j..] += coef (complex accumulate of H[j,k]); tiny
__global__ void k_zacc(int n2, double* __restrict__ dst, const double* __restrict__ src) {
    for (int i = threadIdx.x; i < n2; i += blockDim.x) dst[i] += src[i];
}

static inline int zgrid(kh_ctx ctx, int64_t n) {
    int64_t need = (n + BS - 1) / BS;
    if (need < 1) need = 1;
    return (int)std::min<int64_t>(need, ctx->nb);
}

static int zcheck(kh_vec v, int64_t col, int64_t ncols, const char* what) {
    KH_ARG(v != nullptr, "%s: NULL vector handle", what);
    KH_ARG((v->n & 1) == 0, "%s: a complex vector is a real block of even length", what);
    KH_ARG(col >= 0 && ncols >= 0 && col + ncols <= v->ncols, "%s: columns out of range", what);
    return 0;
}

static inline const double2* zcol(kh_vec v, int64_t j) { return reinterpret_cast<const double2*>(v->col(j)); }
static inline double2* zcolw(kh_vec v, int64_t j) { return reinterpret_cast<double2*>(v->col(j)); }

// out_dev[2c], out_dev[2c+1] = <V_c, w> for c < nc (all-reduced when sharded)
static int zdot_dev(kh_ctx ctx, kh_vec V, int64_t j0, int64_t nc, const double2* w, double* out_dev) {
    const int64_t n = V->n / 2;
    const int grid = zgrid(ctx, n);
    int64_t done = 0;
    while (done < nc) {
        const int64_t left = nc - done;
        const int c = left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
        ZColPtrs cp;
        for (int i = 0; i < c; ++i) cp.c[i] = zcol(V, j0 + done + i);
        switch (c) {
            case 8: hipLaunchKernelGGL((k_zmultidot<8>), dim3(grid), dim3(BS), 0, ctx->stream, n, cp, w, ctx->part, NB_MAX); break;
            case 4: hipLaunchKernelGGL((k_zmultidot<4>), dim3(grid), dim3(BS), 0, ctx->stream, n, cp, w, ctx->part, NB_MAX); break;
            case 2: hipLaunchKernelGGL((k_zmultidot<2>), dim3(grid), dim3(BS), 0, ctx->stream, n, cp, w, ctx->part, NB_MAX); break;
            default: hipLaunchKernelGGL((k_zmultidot<1>), dim3(grid), dim3(BS), 0, ctx->stream, n, cp, w, ctx->part, NB_MAX); break;
        }
        hipLaunchKernelGGL(k_reduce_partials, dim3(2 * c), dim3(BS), 0, ctx->stream, ctx->part, grid, NB_MAX,
                           out_dev + 2 * done, 0);
        KH_HIP(hipGetLastError());
        done += c;
    }
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, out_dev, 2 * nc));
    return 0;
}

template <bool NRM, int BETA>
static void zaxpy_launch(int c, int grid, hipStream_t st, int64_t n, const ZColPtrs& cp, const double2* cf,
                         double sign, double beta, double2* w, double* part) {
    switch (c) {
        case 8: hipLaunchKernelGGL((k_zmultiaxpy<8, NRM, BETA>), dim3(grid), dim3(BS), 0, st, n, cp, cf, sign, beta, w, part); break;
        case 4: hipLaunchKernelGGL((k_zmultiaxpy<4, NRM, BETA>), dim3(grid), dim3(BS), 0, st, n, cp, cf, sign, beta, w, part); break;
        case 2: hipLaunchKernelGGL((k_zmultiaxpy<2, NRM, BETA>), dim3(grid), dim3(BS), 0, st, n, cp, cf, sign, beta, w, part); break;
        default: hipLaunchKernelGGL((k_zmultiaxpy<1, NRM, BETA>), dim3(grid), dim3(BS), 0, st, n, cp, cf, sign, beta, w, part); break;
    }
}

// w = beta*w - sign * sum coef_c X_c over nc columns; nrm: fuse <w,w> partials into the last chunk
static int zaxpy_dev(kh_ctx ctx, kh_vec X, int64_t j0, int64_t nc, const double* coef_dev, double sign, double beta,
                     double2* w, bool nrm, double* nrm_part) {
    const int64_t n = X->n / 2;
    const int grid = zgrid(ctx, n);
    int64_t done = 0;
    while (done < nc) {
        const int64_t left = nc - done;
        const int c = left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
        ZColPtrs cp;
        for (int i = 0; i < c; ++i) cp.c[i] = zcol(X, j0 + done + i);
        const bool last = (done + c == nc);
        const bool tn = nrm && last;
        const int b = (done > 0) ? 1 : (beta == 1.0 ? 1 : (beta == 0.0 ? 0 : 2));
        const double2* cf = reinterpret_cast<const double2*>(coef_dev) + done;
        if (tn) {
            if (b == 1) zaxpy_launch<true, 1>(c, grid, ctx->stream, n, cp, cf, sign, beta, w, nrm_part);
            else if (b == 0) zaxpy_launch<true, 0>(c, grid, ctx->stream, n, cp, cf, sign, beta, w, nrm_part);
            else zaxpy_launch<true, 2>(c, grid, ctx->stream, n, cp, cf, sign, beta, w, nrm_part);
        } else {
            if (b == 1) zaxpy_launch<false, 1>(c, grid, ctx->stream, n, cp, cf, sign, beta, w, nrm_part);
            else if (b == 0) zaxpy_launch<false, 0>(c, grid, ctx->stream, n, cp, cf, sign, beta, w, nrm_part);
            else zaxpy_launch<false, 2>(c, grid, ctx->stream, n, cp, cf, sign, beta, w, nrm_part);
        }
        KH_HIP(hipGetLastError());
        done += c;
    }
    return 0;
}

static int zfetch(kh_ctx ctx, const double* dev, int64_t count, double* out) {
    KH_HIP(hipMemcpyAsync(ctx->hpin, dev, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    KH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->hpin, count * sizeof(double));
    return 0;
}

static int zpush(kh_ctx ctx, const double* host, int64_t count, double* dev) {
    KH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(ctx->hpin, host, count * sizeof(double));
    KH_HIP(hipMemcpyAsync(dev, ctx->hpin, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

constexpr int ZSC_COEF = 6400;   // same scratch region as the real panel coefficients
constexpr int ZSC_TMP = 6144;

// y = A x for complex operators (real kh_vec views of length 2n)
static int zapply_one(kh_ctx ctx, kh_mat A, const double* x, double* y) {
    const double2* x2 = reinterpret_cast<const double2*>(x);
    double2* y2 = reinterpret_cast<double2*>(y);
    if (A->kind == KH_MAT_ZCSR) {
        // a block-row shard: the neighbours' boundary entries first ((re, im) pairs: twice the doubles)
        if (kh_multi(ctx) && (A->nrecv_prev + A->nrecv_next + A->nsend_prev + A->nsend_next) > 0)
            KH_TRY(comm_halo_exchange(ctx, A, x, ctx->stream, 2));
        if (A->nblk == 0) return 0;
        if (A->zdia != nullptr && ctx->spmv_dia && A->nrecv_prev + A->nrecv_next == 0 && x != y) {
            // a banded operator without ghost columns: the diagonal-major copy, no index stream (the same bits)
            DiaOffs o;
            o.nd = A->dia_nd;
            for (int d = 0; d < KH_DIA_MAX; ++d) o.off[d] = d < A->dia_nd ? A->dia_off[d] : 0;
            const double2* zd = reinterpret_cast<const double2*>(A->zdia);
            const int grid = zgrid(ctx, A->n_rows);
            if (o.nd == 5) hipLaunchKernelGGL((k_zspmv_dia<5>), dim3(grid), dim3(BS), 0, ctx->stream, o, zd, A->zdia_ld, A->n_rows, x2, y2);
            else if (o.nd == 7) hipLaunchKernelGGL((k_zspmv_dia<7>), dim3(grid), dim3(BS), 0, ctx->stream, o, zd, A->zdia_ld, A->n_rows, x2, y2);
            else hipLaunchKernelGGL((k_zspmv_dia<0>), dim3(grid), dim3(BS), 0, ctx->stream, o, zd, A->zdia_ld, A->n_rows, x2, y2);
            KH_HIP(hipGetLastError());
            ctx->n_zspmv_dia += 1;
            return 0;
        }
        const size_t lds = (size_t)A->tile * sizeof(double2);
        const double2* d2 = reinterpret_cast<const double2*>(A->data);
        const int64_t nloc = A->n_cols - A->nrecv_prev - A->nrecv_next;
        const double2* gh = reinterpret_cast<const double2*>(A->ghost);
        switch (A->tile / BS) {
            case 4: hipLaunchKernelGGL((k_zspmv_stream<4>), dim3(A->nblk), dim3(BS), lds, ctx->stream, A->indptr, A->indices, d2, A->rowblk, A->nblk, A->tile, x2, y2, nloc, gh); break;
            case 16: hipLaunchKernelGGL((k_zspmv_stream<16>), dim3(A->nblk), dim3(BS), lds, ctx->stream, A->indptr, A->indices, d2, A->rowblk, A->nblk, A->tile, x2, y2, nloc, gh); break;
            default: hipLaunchKernelGGL((k_zspmv_stream<8>), dim3(A->nblk), dim3(BS), lds, ctx->stream, A->indptr, A->indices, d2, A->rowblk, A->nblk, A->tile, x2, y2, nloc, gh); break;
        }
    } else if (A->kind == KH_MAT_ZDENSE) {
        const int grid = (int)((A->n_rows + BS / 64 - 1) / (BS / 64));
        hipLaunchKernelGGL(k_zgemv_dense, dim3(grid), dim3(BS), 0, ctx->stream, A->n_rows, A->n_cols,
                           reinterpret_cast<const double2*>(A->a), A->lda, x2, y2);
    } else {
        hipLaunchKernelGGL(k_zdiag_apply, dim3(zgrid(ctx, A->n_rows)), dim3(BS), 0, ctx->stream, A->n_rows,
                           reinterpret_cast<const double2*>(A->diag), x2, y2);
    }
    KH_HIP(hipGetLastError());
    return 0;
}

}  // namespace kh

using namespace kh;

static int zapply_cols(kh_ctx ctx, kh_mat A, kh_vec X, int64_t xcol, kh_vec Y, int64_t ycol, int64_t ncols) {
    KH_TRY(zcheck(X, xcol, ncols, "kh_apply(X, complex)"));
    KH_TRY(zcheck(Y, ycol, ncols, "kh_apply(Y, complex)"));
    KH_ARG(X->n == 2 * (A->n_cols - A->nrecv_prev - A->nrecv_next) && Y->n == 2 * A->n_rows,
           "kh_apply: dimension mismatch (complex A %lldx%lld, x %lld, y %lld reals)", (long long)A->n_rows,
           (long long)A->n_cols, (long long)X->n, (long long)Y->n);
    KH_ARG(!(X == Y && xcol == ycol), "kh_apply: in-place application is not supported");
    for (int64_t c = 0; c < ncols; ++c) KH_TRY(zapply_one(ctx, A, X->col(xcol + c), Y->col(ycol + c)));
    return 0;
}

extern "C" {

int kh_zcsr_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* indptr,
                   const int32_t* indices, const double* data, kh_mat* out) {
    KH_ARG(ctx && out && indptr, "kh_zcsr_upload: NULL argument");
    KH_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_rows < 2147483647LL && nnz < 2147483647LL,
           "kh_zcsr_upload: bad sizes");
    KH_ARG(indptr[0] == 0 && indptr[n_rows] == nnz, "kh_zcsr_upload: inconsistent indptr");
    for (int64_t i = 0; i < nnz; ++i)
        KH_ARG(indices[i] >= 0 && indices[i] < n_cols, "kh_zcsr_upload: column index out of range");
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_ZCSR;
    A->n_rows = n_rows;
    A->n_cols = n_cols;
    A->nnz = nnz;
    A->tile = std::min(ctx->spmv_tile, 2048);   // double2 products: 32 KB of LDS at 2048
    // row blocks: same greedy rule as the real kernel
    std::vector<int32_t> blk;
    blk.push_back(0);
    int64_t r = 0;
    while (r < n_rows) {
        int64_t r_end = r, acc = 0;
        while (r_end < n_rows && (r_end - r) < BS * 4) {
            const int64_t nz = (int64_t)indptr[r_end + 1] - indptr[r_end];
            if (acc + nz > A->tile) break;
            acc += nz;
            ++r_end;
        }
        if (r_end == r) r_end = r + 1;
        blk.push_back((int32_t)r_end);
        r = r_end;
    }
    A->nblk = (int)blk.size() - 1;
    KH_HIP(hipMalloc(&A->indptr, sizeof(int32_t) * (n_rows + 1)));
    KH_HIP(hipMalloc(&A->indices, sizeof(int32_t) * std::max<int64_t>(nnz, 1)));
    KH_HIP(hipMalloc(&A->data, sizeof(double) * 2 * std::max<int64_t>(nnz, 1)));
    KH_HIP(hipMalloc(&A->rowblk, sizeof(int32_t) * blk.size()));
    KH_HIP(hipMemcpy(A->indptr, indptr, sizeof(int32_t) * (n_rows + 1), hipMemcpyHostToDevice));
    if (nnz > 0) {
        KH_HIP(hipMemcpy(A->indices, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
        KH_HIP(hipMemcpy(A->data, data, sizeof(double) * 2 * nnz, hipMemcpyHostToDevice));
    }
    KH_HIP(hipMemcpy(A->rowblk, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice));
    // a banded complex operator (a shifted stencil matrix) gets a diagonal-major copy like the real ones: the complex chain
    // kernels compute w = A v_k from it in their prologue (chain.h, CPLX + FND); kh_apply keeps the CSR-stream kernel
    if (n_rows == n_cols && nnz > 0 && ctx->spmv_dia && ctx->chain_spmv) {
        std::vector<int> offs;
        if (detect_dia(n_rows, n_cols, nnz, indptr, indices, data, offs, 0, 0, 2) && (offs.size() == 5 || offs.size() == 7)) {
            const int64_t ld = std::max<int64_t>(padded_ld(ctx, 2 * n_rows) / 2, ((n_rows + 31) / 32) * 32);
            double* zd = nullptr;
            if (hipMalloc(&zd, sizeof(double) * 2 * (size_t)ld * offs.size()) == hipSuccess) {
                KH_HIP(hipMemsetAsync(zd, 0, sizeof(double) * 2 * (size_t)ld * offs.size(), ctx->stream));
                DiaOffs o;
                o.nd = (int)offs.size();
                for (int d = 0; d < KH_DIA_MAX; ++d) o.off[d] = d < o.nd ? offs[d] : 0;
                hipLaunchKernelGGL(k_zdia_fill, dim3((unsigned)((n_rows + BS - 1) / BS)), dim3(BS), 0, ctx->stream, A->indptr,
                                   A->indices, reinterpret_cast<const double2*>(A->data), n_rows, o,
                                   reinterpret_cast<double2*>(zd), ld);
                KH_HIP(hipGetLastError());
                KH_HIP(hipStreamSynchronize(ctx->stream));
                A->zdia = zd;
                A->zdia_ld = ld;
                A->dia_nd = o.nd;
                for (int d = 0; d < o.nd; ++d) A->dia_off[d] = offs[d];
            } else {
                (void)hipGetLastError();
            }
        }
    }
    *out = A;
    return 0;
}

int kh_zdense_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, const double* a, int64_t lda, kh_mat* out) {
    KH_ARG(ctx && out && a && lda >= n_cols, "kh_zdense_upload: bad argument");
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_ZDENSE;
    A->n_rows = n_rows;
    A->n_cols = n_cols;
    A->lda = n_cols;
    KH_HIP(hipMalloc(&A->a, sizeof(double) * 2 * (size_t)std::max<int64_t>(n_rows * n_cols, 1)));
    KH_HIP(hipMemcpy2D(A->a, A->lda * 16, a, lda * 16, n_cols * 16, n_rows, hipMemcpyHostToDevice));
    *out = A;
    return 0;
}

int kh_zdiag_upload(kh_ctx ctx, int64_t n, const double* d, kh_mat* out) {
    KH_ARG(ctx && out && (d || n == 0), "kh_zdiag_upload: NULL argument");
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_ZDIAG;
    A->n_rows = A->n_cols = n;
    KH_HIP(hipMalloc(&A->diag, sizeof(double) * 2 * std::max<int64_t>(n, 1)));
    if (n > 0) KH_HIP(hipMemcpy(A->diag, d, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    *out = A;
    return 0;
}

int kh_zfrom_real(kh_ctx ctx, kh_vec X, int64_t xcol, kh_vec Z, int64_t zcol_, int64_t ncols) {
    KH_ARG(ctx && X, "kh_zfrom_real: NULL");
    KH_ARG(xcol >= 0 && ncols >= 0 && xcol + ncols <= X->ncols, "kh_zfrom_real: columns out of range");
    KH_TRY(zcheck(Z, zcol_, ncols, "kh_zfrom_real(Z)"));
    KH_ARG(Z->n == 2 * X->n, "kh_zfrom_real: Z must hold 2*len(X) reals");
    for (int64_t c = 0; c < ncols; ++c)
        hipLaunchKernelGGL(k_zfrom_real, dim3(zgrid(ctx, X->n)), dim3(BS), 0, ctx->stream, X->n, X->col(xcol + c),
                           zcolw(Z, zcol_ + c));
    KH_HIP(hipGetLastError());
    return 0;
}

int kh_zdot_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, kh_vec W, int64_t wcol, double* out) {
    KH_ARG(ctx && out, "kh_zdot_panel: NULL");
    KH_TRY(zcheck(V, j0, ncols, "kh_zdot_panel(V)"));
    KH_TRY(zcheck(W, wcol, 1, "kh_zdot_panel(W)"));
    KH_ARG(V->n == W->n && ncols <= 512, "kh_zdot_panel: shapes");
    if (ncols == 0) return 0;
    double* dev = ctx->scal + ZSC_COEF;
    KH_TRY(zdot_dev(ctx, V, j0, ncols, zcol(W, wcol), dev));
    return zfetch(ctx, dev, 2 * ncols, out);
}

int kh_zaxpy_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* h, kh_vec W, int64_t wcol) {
    KH_ARG(ctx && (h || ncols == 0), "kh_zaxpy_panel: NULL");
    KH_TRY(zcheck(V, j0, ncols, "kh_zaxpy_panel(V)"));
    KH_TRY(zcheck(W, wcol, 1, "kh_zaxpy_panel(W)"));
    KH_ARG(V->n == W->n && ncols <= 512, "kh_zaxpy_panel: shapes");
    if (ncols == 0) return 0;
    double* dev = ctx->scal + ZSC_COEF;
    KH_TRY(zpush(ctx, h, 2 * ncols, dev));
    return zaxpy_dev(ctx, V, j0, ncols, dev, 1.0, 1.0, zcolw(W, wcol), false, nullptr);
}

int kh_zgemm_nn(kh_ctx ctx, kh_vec X, int64_t x0, int64_t k, const double* C, int64_t nc, double beta, kh_vec Y,
                int64_t y0) {
    KH_ARG(ctx && (C || k == 0 || nc == 0), "kh_zgemm_nn: NULL");
    KH_TRY(zcheck(X, x0, k, "kh_zgemm_nn(X)"));
    KH_TRY(zcheck(Y, y0, nc, "kh_zgemm_nn(Y)"));
    KH_ARG(X->n == Y->n && k <= 512 && X != Y, "kh_zgemm_nn: shapes");
    std::vector<double> coef((size_t)std::max<int64_t>(2 * k, 2));
    double* dev = ctx->scal + ZSC_COEF;
    for (int64_t c = 0; c < nc; ++c) {
        if (k == 0) {
            if (beta == 0.0) KH_TRY(kh_vec_zero(Y, y0 + c, 1));
            continue;
        }
        for (int64_t i = 0; i < k; ++i) {
            coef[2 * i] = C[2 * (i * nc + c)];
            coef[2 * i + 1] = C[2 * (i * nc + c) + 1];
        }
        KH_TRY(zpush(ctx, coef.data(), 2 * k, dev));
        KH_TRY(zaxpy_dev(ctx, X, x0, k, dev, -1.0, beta, zcolw(Y, y0 + c), false, nullptr));
    }
    return 0;
}

int kh_zwaxpby(kh_ctx ctx, kh_vec Z, int64_t zcol_, const double alpha[2], kh_vec X, int64_t xcol,
               const double beta[2], kh_vec Y, int64_t ycol) {
    KH_ARG(ctx && alpha && beta, "kh_zwaxpby: NULL");
    KH_TRY(zcheck(Z, zcol_, 1, "kh_zwaxpby(Z)"));
    KH_TRY(zcheck(X, xcol, 1, "kh_zwaxpby(X)"));
    KH_TRY(zcheck(Y, ycol, 1, "kh_zwaxpby(Y)"));
    KH_ARG(Z->n == X->n && Z->n == Y->n, "kh_zwaxpby: length mismatch");
    const int64_t n = Z->n / 2;
    double2 a, b;
    a.x = alpha[0];
    a.y = alpha[1];
    b.x = beta[0];
    b.y = beta[1];
    hipLaunchKernelGGL(k_zwaxpby, dim3(zgrid(ctx, n)), dim3(BS), 0, ctx->stream, n, zcolw(Z, zcol_), a,
                       zcol(X, xcol), b, zcol(Y, ycol));
    KH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

// One complex Arnoldi.advance() (utils.py:954-1048, mgs/dmgs/lanczos/cgs; no preconditioner), enqueued
// on the context's stream: the H column (k+2 complex numbers, the last one (H[k+1,k], 0)) lands in
// `hdev`.  h_km1_dev: Lanczos coefficient still on the device (look-ahead), else h_km1 from the host.
// y = M x for a tiny dense complex row-major M (d x d) on the device: the projector's R^{-1} Q^H and WR^H.
// One workgroup, one row per thread, NumPy's complex multiply, sums left to right (deterministic).
__global__ __launch_bounds__(BS) void k_zsmall_matvec(int d, const double2* __restrict__ M, const double2* __restrict__ x,
                                                      double2* __restrict__ y) {
    for (int i = threadIdx.x; i < d; i += BS) {
        double2 s = make_double2(0.0, 0.0);
        if (M == nullptr) {
            s = x[i];
        } else {
            for (int j = 0; j < d; ++j) {
                const double2 m = M[(int64_t)i * d + j], v = x[j];
                s.x += m.x * v.x - m.y * v.y;
                s.y += m.x * v.y + m.y * v.x;
            }
        }
        y[i] = s;
    }
}

// complex deflation projector on the device (the c128 twin of proj_apply_dev): z <- complement projection of z,
// ya_dev (d complex numbers) gets <Y, z_in>
static int zproj_apply_dev(kh_ctx ctx, kh_proj p, double2* z, double* ya_dev) {
    const int d = (int)p->d;
    for (int it = 0; it < p->iterations; ++it) {
        KH_TRY(zdot_dev(ctx, p->W, 0, d, z, p->c0));          // (all-reduced over the ranks inside)
        if (it == 0 && ya_dev != nullptr)
            hipLaunchKernelGGL(k_zsmall_matvec, dim3(1), dim3(BS), 0, ctx->stream, d,
                               reinterpret_cast<const double2*>(p->WRH), reinterpret_cast<const double2*>(p->c0),
                               reinterpret_cast<double2*>(ya_dev));
        hipLaunchKernelGGL(k_zsmall_matvec, dim3(1), dim3(BS), 0, ctx->stream, d,
                           reinterpret_cast<const double2*>(p->T), reinterpret_cast<const double2*>(p->c0),
                           reinterpret_cast<double2*>(p->c1));
        KH_HIP(hipGetLastError());
        KH_TRY(zaxpy_dev(ctx, p->V, 0, d, p->c1, 1.0, 1.0, z, false, nullptr));
    }
    return 0;
}

static int zstep_enqueue(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t wcol, int64_t k, int64_t start,
                         int sweeps, int gs_mode, const double h_km1[2], const double* h_km1_dev, double* hdev,
                         int slot, kh_proj proj = nullptr, kh_mat Md = nullptr, kh_vec P = nullptr) {
    KH_TRY(zcheck(V, k, 2, "kh_zarnoldi_step(V)"));
    KH_TRY(zcheck(W, wcol, Md ? 2 : 1, "kh_zarnoldi_step(W)"));
    // Md: the Jacobi preconditioner as a complex diagonal, V = Md P (utils.py:1026-1045): dots against V, updates
    // with P, norm sqrt(Re <w, Md w>), both blocks get their next column
    KH_ARG((Md == nullptr) == (P == nullptr), "kh_zarnoldi_step: P and Md go together");
    // (Md may also be a complex CSR / dense matrix - a preconditioner given as a matrix: the tail applies it)
    KH_ARG(Md == nullptr || (Md->kind >= KH_MAT_ZCSR && 2 * Md->n_rows == V->n && Md->n_cols == Md->n_rows &&
                             P->n == V->n && P->ncols >= V->ncols),
           "kh_zarnoldi_step: Md must be a complex operator of the vectors' length, P a block like V");
    kh_vec Bk = P ? P : V;
    KH_ARG(V->n == W->n && start >= 0 && start <= k && sweeps >= 1 && sweeps <= 4, "kh_zarnoldi_step: arguments");
    const int64_t n = V->n / 2;
    double* tmp = ctx->scal + ZSC_TMP;
    double* coef = ctx->scal + ZSC_COEF;
    double2* w = zcolw(W, wcol);
    KH_HIP(hipMemsetAsync(hdev, 0, sizeof(double) * 2 * (k + 2), ctx->stream));
    if (A != nullptr) {
        KH_ARG(A->kind >= KH_MAT_ZCSR && A->n_rows == n, "kh_zarnoldi_step: complex operator of matching size needed");
        // a banded complex operator, reference-order Gram-Schmidt, no projector / preconditioner / Lanczos pre-subtraction:
        // the chain kernel computes w = A v_k in its prologue (no SpMV launch, w never written to memory)
        // (a Lanczos step too: its pre-subtraction w -= H[k,k-1] v_{k-1} has a REAL coefficient - the norm of the previous
        // step - and runs inside the kernel like the real one's; a coefficient with an imaginary part takes the launches)
        const bool lz = (start > 0 && start == k);
        if (gs_mode == KH_GS_MGS && Md == nullptr && proj == nullptr && A->kind == KH_MAT_ZCSR && A->zdia != nullptr &&
            (!lz || h_km1_dev != nullptr || h_km1[1] == 0.0)) {
            const int rc = try_chain(ctx, V, V, W->col(wcol), W->ld, nullptr, nullptr, k, start, sweeps, lz, lz ? h_km1[0] : 0.0,
                                     lz ? h_km1_dev : nullptr, hdev, slot, true, nullptr, 0, A, V->col(k));
            if (rc < 0) return rc;
            if (rc == 1) return 0;
        }
        KH_TRY(zapply_one(ctx, A, V->col(k), W->col(wcol)));
        // deflated solvers: w <- (I - P) w, and <U, A v_k> behind the H column (deflation.py:135-143)
        if (proj != nullptr) KH_TRY(zproj_apply_dev(ctx, proj, w, hdev + 2 * (k + 2)));
    }
    double* nrm_part = ctx->part + (int64_t)(2 * ZMAXC + 2) * NB_MAX;
    const int grid = zgrid(ctx, n);
    if (start > 0 && start == k) {   // Lanczos: w -= H[k,k-1] * v_{k-1}
        const double* cf = h_km1_dev;
        if (cf == nullptr) {
            KH_TRY(zpush(ctx, h_km1, 2, coef));
            cf = coef;
        }
        KH_TRY(zaxpy_dev(ctx, Bk, k - 1, 1, cf, 1.0, 1.0, w, false, nullptr));
    }
    const int64_t ncol = k - start + 1;
    // reference-order MGS with w in registers for the whole chain (chain.h, CPLX instantiation):
    // one launch instead of 4 per column
    if (gs_mode == KH_GS_MGS && Md == nullptr) {
        const int rc = try_chain(ctx, V, V, W->col(wcol), W->ld, nullptr, nullptr, k, start, sweeps, false, 0.0,
                                 nullptr, hdev, slot, true);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    // panel (classical) Gram-Schmidt with w register-resident: two launches per sweep (chain.h, CPLX instantiations)
    const double* nrm_src = nrm_part;
    int nrm_n = grid;
    bool panel_done = false;
    if (gs_mode != KH_GS_MGS && Md == nullptr) {
        int cnt = 0;
        const int rc = try_cgs_reg(ctx, V, V, W->col(wcol), W->ld, nullptr, nullptr, start, ncol, sweeps, kh_multi(ctx),
                                   hdev, coef, &cnt, true);
        if (rc < 0) return rc;
        if (rc == 1) {
            panel_done = true;
            nrm_src = part_slot(ctx, SLOT_NRM);
            nrm_n = cnt;
        }
    }
    for (int s = 0; s < sweeps && !panel_done; ++s) {
        const bool last_sweep = (s == sweeps - 1);
        if (gs_mode == KH_GS_MGS) {
            for (int64_t j = start; j <= k; ++j) {
                KH_TRY(zdot_dev(ctx, V, j, 1, w, coef));
                hipLaunchKernelGGL(k_zacc, dim3(1), dim3(64), 0, ctx->stream, 2, hdev + 2 * j, coef);
                KH_TRY(zaxpy_dev(ctx, Bk, j, 1, coef, 1.0, 1.0, w, !Md && last_sweep && j == k, nrm_part));
            }
        } else {
            KH_ARG(ncol <= 512, "kh_zarnoldi_step: panel mode handles at most 512 columns");
            KH_TRY(zdot_dev(ctx, V, start, ncol, w, coef));
            hipLaunchKernelGGL(k_zacc, dim3(1), dim3(256), 0, ctx->stream, (int)(2 * ncol), hdev + 2 * start, coef);
            KH_TRY(zaxpy_dev(ctx, Bk, start, ncol, coef, 1.0, 1.0, w, !Md && last_sweep, nrm_part));
        }
    }
    // norm (real) and normalise through the real kernels on the 2n view
    const int rgrid = (int)std::min<int64_t>(std::max<int64_t>(((V->n >> 1) + BS - 1) / BS, 1), ctx->nb);
    if (Md != nullptr) {
        // mw = Md w; H[k+1,k] = sqrt(Re <w, mw>) (the real dot of the two 2n views); p_{k+1} = w / h, v_{k+1} = mw / h
        KH_TRY(zapply_one(ctx, Md, W->col(wcol), W->col(wcol + 1)));
        KH_TRY(dot_panel_dev(ctx, W, wcol, 1, W->col(wcol + 1), tmp, 0));
        if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
        hipLaunchKernelGGL((k_scale_store<A_SCAL>), dim3(rgrid), dim3(BS), 0, ctx->stream, V->n, W->col(wcol),
                           W->col(wcol + 1), V->col(k + 1), P->col(k + 1), nullptr, 0, tmp, hdev + 2 * (k + 1));
        KH_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, nrm_src, nrm_n, 0, tmp, 0);
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
    hipLaunchKernelGGL((k_scale_store<A_SCAL>), dim3(rgrid), dim3(BS), 0, ctx->stream, V->n, W->col(wcol), nullptr,
                       V->col(k + 1), nullptr, nullptr, 0, tmp, hdev + 2 * (k + 1));
    KH_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int kh_zarnoldi_step(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t wcol, int64_t k, int64_t start,
                     int sweeps, int gs_mode, const double h_km1[2], double* hcol_out) {
    KH_ARG(ctx && V && W && hcol_out && h_km1, "kh_zarnoldi_step: NULL argument");
    KH_ARG(2 * (k + 2) <= 4096, "kh_zarnoldi_step: k too large for the synchronous complex step");
    double* hdev = ctx->scal;                 // 2(k+2) doubles
    KH_TRY(zstep_enqueue(ctx, A, V, W, wcol, k, start, sweeps, gs_mode, h_km1, nullptr, hdev, 0));
    KH_TRY(zfetch(ctx, hdev, 2 * (k + 2), hcol_out));
    if (*ctx->chain_err_pin[0] != 0) {
        *ctx->chain_err_pin[0] = 0;
        chain_switch_off(ctx);
        (void)hipMemsetAsync(ctx->chain_err, 0, sizeof(int), ctx->stream);
        return fail(KH_ERR_HIP, "grid-wide reduction of the complex MGS chain kernel timed out; the chain "
                                "path is now disabled for this context");
    }
    return 0;
}

// The same step split like kh_arnoldi_step_begin / _end (look-ahead): results are collected with
// kh_arnoldi_step_end(ctx, slot, 2*(k+2), out).  h_km1[0] = NaN: Lanczos coefficient H[k,k-1] of the
// step begun just before this one, still in the previous H-column slot on the device.
int kh_zarnoldi_step_begin_md(kh_ctx ctx, kh_mat A, kh_proj proj, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t wcol,
                              int64_t k, int64_t start, int sweeps, int gs_mode, const double h_km1[2], int slot) {
    KH_ARG(ctx && V && W && h_km1, "kh_zarnoldi_step_begin: NULL argument");
    KH_ARG(slot >= 0 && slot < KH_NSLOT, "kh_zarnoldi_step_begin: slot %d not in [0,%d)", slot, KH_NSLOT);
    KH_ARG(k >= 0 && k + 1 < V->ncols, "kh_zarnoldi_step_begin: k=%lld needs %lld basis columns, have %lld",
           (long long)k, (long long)(k + 2), (long long)V->ncols);
    const int64_t pd = proj ? proj->d : 0;
    KH_ARG(proj == nullptr || (proj->cplx && proj->W->n == V->n && A != nullptr),
           "kh_zarnoldi_step_begin: a complex projector of length N and the operator are needed");
    KH_TRY(ensure_hcap(ctx, 2 * (std::max<int64_t>(k + 2, V->ncols + 1) + pd)));
    ctx->wait_tag[slot] = false;
    chain_rearm(ctx, k);
    {
        kh_step_s& st = ctx->step[slot];
        st.kind = 2;
        st.A = A; st.proj = proj; st.Md = Md; st.V = V; st.P = P; st.W = W;
        st.wcol = wcol; st.k = k; st.start = start; st.sweeps = sweeps; st.gs_mode = gs_mode;
        st.h_km1[0] = h_km1[0]; st.h_km1[1] = h_km1[1];
        if (step_poisoned(ctx, slot)) {
            KH_HIP(hipEventRecord(ctx->hev[slot], ctx->stream));
            return 0;
        }
    }
    double* hdev = ctx->hslot_dev[slot];
    const double* hk_dev = nullptr;
    if (start > 0 && start == k && h_km1[0] != h_km1[0])
        hk_dev = ctx->hslot_dev[(slot + KH_NSLOT - 1) % KH_NSLOT] + 2 * k;
    KH_TRY(zstep_enqueue(ctx, A, V, W, wcol, k, start, sweeps, gs_mode, h_km1, hk_dev, hdev, slot, proj, Md, P));
    KH_HIP(hipMemcpyAsync(ctx->hslot_pin[slot], hdev, sizeof(double) * 2 * (k + 2 + pd), hipMemcpyDeviceToHost,
                          ctx->stream));
    KH_HIP(hipEventRecord(ctx->hev[slot], ctx->stream));
    return 0;
}

int kh_zarnoldi_step_begin_proj(kh_ctx ctx, kh_mat A, kh_proj proj, kh_vec V, kh_vec W, int64_t wcol, int64_t k,
                                int64_t start, int sweeps, int gs_mode, const double h_km1[2], int slot) {
    return kh_zarnoldi_step_begin_md(ctx, A, proj, nullptr, V, nullptr, W, wcol, k, start, sweeps, gs_mode, h_km1, slot);
}

int kh_zarnoldi_step_begin(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t wcol, int64_t k, int64_t start,
                           int sweeps, int gs_mode, const double h_km1[2], int slot) {
    return kh_zarnoldi_step_begin_md(ctx, A, nullptr, nullptr, V, nullptr, W, wcol, k, start, sweeps, gs_mode, h_km1, slot);
}

// Complex MINRES update in one pass (linsys.py:844-846); coefficients as (re, im) pairs.
int kh_zminres_update(kh_ctx ctx, kh_vec V, int64_t k, kh_vec Wk, int slot, const double r0[2], const double r1[2],
                      const double r2[2], const double y0[2], kh_vec YK, int64_t ycol) {
    KH_ARG(ctx && r0 && r1 && r2 && y0, "kh_zminres_update: NULL");
    KH_TRY(zcheck(V, k, 1, "kh_zminres_update(V)"));
    KH_TRY(zcheck(Wk, 0, 2, "kh_zminres_update(W)"));
    KH_TRY(zcheck(YK, ycol, 1, "kh_zminres_update(yk)"));
    KH_ARG(slot == 0 || slot == 1, "kh_zminres_update: slot");
    KH_ARG(V->n == Wk->n && V->n == YK->n, "kh_zminres_update: length mismatch");
    KH_ARG(!(YK == Wk) && !(YK == V && ycol == k) && !(V == Wk), "kh_zminres_update: yk, v_k and W must not overlap");
    const int64_t n = V->n / 2;
    auto c2 = [](const double* q) { double2 r; r.x = q[0]; r.y = q[1]; return r; };
    hipLaunchKernelGGL(k_zminres_update, dim3(zgrid(ctx, n)), dim3(BS), 0, ctx->stream, n, zcol(V, k), zcolw(Wk, slot),
                       zcol(Wk, 1 - slot), c2(r0), c2(r1), c2(r2), c2(y0), zcolw(YK, ycol));
    KH_HIP(hipGetLastError());
    return 0;
}

// One whole pass of the CG loop (linsys.py:622-665) for complex data with a single host synchronisation: the
// complex twin of kh_cg_step.  A: complex operator; the vectors are real blocks of length 2N (interleaved re, im);
// Md: NULL or a REAL diagonal of length 2N (every entry of the real Jacobi scaling twice) - the recurrences have real
// coefficients, so they run on the real views.  out[0] = d with alpha = rho / d = Re(rho / <p, Ap>) (what the device
// used), out[1] = <r, z> (real part), out[2], out[3] = <p, Ap>.
int kh_zcg_step(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol, kh_vec YK,
                int64_t ycol, kh_vec R, int64_t rcol, kh_vec Z, int64_t zcol_, int first, double omega, double rho,
                double* out) {
    KH_ARG(ctx && A && out, "kh_zcg_step: NULL");
    RoctxScope range_(ctx, "kh_zcg_step");
    KH_TRY(zcheck(Pd, pcol, 1, "kh_zcg_step(p)"));
    KH_TRY(zcheck(AP, apcol, 1, "kh_zcg_step(Ap)"));
    KH_TRY(zcheck(YK, ycol, 1, "kh_zcg_step(yk)"));
    KH_TRY(zcheck(R, rcol, 1, "kh_zcg_step(r)"));
    KH_ARG(A->kind >= KH_MAT_ZCSR, "kh_zcg_step: complex operator expected");
    const int64_t n = R->n;              // reals
    KH_ARG(Md == nullptr || (Md->kind == KH_MAT_DIAG && Md->n_rows == n),
           "kh_zcg_step: Md must be a real diagonal of length 2N (entries duplicated)");
    if (Md) KH_TRY(zcheck(Z, zcol_, 1, "kh_zcg_step(z)"));
    KH_ARG(Pd->n == n && AP->n == n && YK->n == n && 2 * A->n_rows == n && (!Md || Z->n == n),
           "kh_zcg_step: length mismatch");
    KH_ARG(!(Pd == AP && pcol == apcol), "kh_zcg_step: p and Ap must be different columns");
    double* p = Pd->col(pcol);
    double* ap = AP->col(apcol);
    double* r = R->col(rcol);
    double* z = Md ? Z->col(zcol_) : r;
    const int grid = grid_lin(ctx, n);
    double* tmp = ctx->scal + SC_TMP;       // tmp[0] = d, tmp[1] = rho_new, tmp[2..3] = <p, Ap>
    if (!first)                             // p = z + omega p   (linsys.py:627)
        hipLaunchKernelGGL(k_waxpby, dim3(grid), dim3(BS), 0, ctx->stream, n, p, 1.0, z, omega, p);
    KH_TRY(zapply_one(ctx, A, p, ap));
    KH_TRY(zdot_dev(ctx, Pd, pcol, 1, reinterpret_cast<const double2*>(ap), tmp + 2));
    hipLaunchKernelGGL(k_zcg_den, dim3(1), dim3(1), 0, ctx->stream, tmp);
    double* part = part_slot(ctx, SLOT_NRM);
    if (Md)
        hipLaunchKernelGGL((k_cg_update<true>), dim3(grid), dim3(BS), 0, ctx->stream, n, rho, p, ap,
                           YK->col(ycol), r, Md->diag, z, part, tmp);
    else
        hipLaunchKernelGGL((k_cg_update<false>), dim3(grid), dim3(BS), 0, ctx->stream, n, rho, p, ap,
                           YK->col(ycol), r, nullptr, nullptr, part, tmp);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, part, grid, 0, tmp + 1, 0);
    KH_HIP(hipGetLastError());
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp + 1, 1));
    KH_TRY(fetch_scalars(ctx, tmp, 4, out));
    out[4] = (double)(cg_sanity(out[0], out[1], rho) | ((std::isfinite(out[2]) && std::isfinite(out[3])) ? 0 : KH_CG_NONFINITE_PAP));
    return 0;
}

// complex projector: W, V complex blocks (real kh_vec of length 2N), T = R^{-1} Q^H and WRH = WR^H as d x d
// row-major (re, im) pairs (NULL: identity)
int kh_zproj_create(kh_ctx ctx, kh_vec W, kh_vec V, int64_t d, const double* T, const double* WRH, int iterations,
                    kh_proj* out) {
    KH_ARG(ctx && W && V && out, "kh_zproj_create: NULL");
    KH_ARG(d >= 1 && d <= ZMAXD && W->ncols >= d && V->ncols >= d && W->n == V->n && (W->n & 1) == 0,
           "kh_zproj_create: shapes");
    KH_ARG(iterations >= 1, "kh_zproj_create: iterations < 1");
    kh_proj p = new kh_proj_s();
    p->ctx = ctx;
    p->W = W;
    p->V = V;
    p->d = d;
    p->iterations = iterations;
    p->cplx = 1;
    auto body = [&]() -> int {
        KH_HIP(hipMalloc(&p->c0, sizeof(double) * 6 * d));
        p->c1 = p->c0 + 2 * d;
        p->ya = p->c0 + 4 * d;
        if (T) {
            KH_HIP(hipMalloc(&p->T, sizeof(double) * 2 * d * d));
            KH_HIP(hipMemcpy(p->T, T, sizeof(double) * 2 * d * d, hipMemcpyHostToDevice));
        }
        if (WRH) {
            KH_HIP(hipMalloc(&p->WRH, sizeof(double) * 2 * d * d));
            KH_HIP(hipMemcpy(p->WRH, WRH, sizeof(double) * 2 * d * d, hipMemcpyHostToDevice));
        }
        return 0;
    };
    const int rc = body();
    if (rc != 0) {
        kh_proj_free(p);
        return rc;
    }
    *out = p;
    return 0;
}

int kh_zproj_apply_complement(kh_ctx ctx, kh_proj p, kh_vec A, int64_t acol, kh_vec Z, int64_t zcol_, double* ya_out) {
    KH_ARG(ctx && p && p->cplx, "kh_zproj_apply_complement: complex projector needed");
    KH_TRY(zcheck(A, acol, 1, "kh_zproj_apply_complement(a)"));
    KH_TRY(zcheck(Z, zcol_, 1, "kh_zproj_apply_complement(z)"));
    KH_ARG(A->n == p->W->n && Z->n == A->n, "kh_zproj_apply_complement: length mismatch");
    if (!(A == Z && acol == zcol_))
        KH_HIP(hipMemcpyAsync(Z->col(zcol_), A->col(acol), sizeof(double) * A->n, hipMemcpyDeviceToDevice,
                              ctx->stream));
    KH_TRY(zproj_apply_dev(ctx, p, zcolw(Z, zcol_), ya_out ? p->ya : nullptr));
    if (ya_out) return zfetch(ctx, p->ya, 2 * p->d, ya_out);
    return 0;
}

}  // extern "C"
