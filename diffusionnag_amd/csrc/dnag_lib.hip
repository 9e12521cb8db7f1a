// Host side of the dnag C ABI (include/dnag.h): weight packing, workspace carving, kernel launches
// and the predictor-corrector loop.  Built with hipcc --offload-arch=gfx950 into libdnag.so.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>

#include "dnag_common.h"
#include "dnag_encoder.hip.h"
#include "dnag_encoder8.hip.h"
#include "dnag_head.hip.h"
#include "dnag_gemm16.hip.h"
#include "dnag_gemm_dma.hip.h"
#include "dnag_predictor.hip.h"
#include "dnag_predcate.hip.h"
#include "dnag_train.hip.h"
#include "dnag_arch.hip.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(DNAG_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int dnag_abi_version(void) { return 1; }
extern "C" size_t dnag_pc_args_bytes(void) { return sizeof(dnag_pc_args); }
extern "C" size_t dnag_train_args_bytes(void) { return sizeof(dnag_train_args); }
extern "C" const char *dnag_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// shape checks
// ---------------------------------------------------------------------------------------------
static bool is_nb201(const dnag_cate_dims &d) { return d.max_node == 8 && d.n_vocab == 7; }
static bool is_ofa(const dnag_cate_dims &d) { return d.max_node == 20 && d.n_vocab == 9; }

static int check_cate(const dnag_cate_dims *d) {
    if (!d) return fail(DNAG_E_ARG, "cate dims is NULL");
    if (d->d_model != 64 || d->n_head != 8 || d->d_ff != 128)
        return fail(DNAG_E_ARG, "unsupported CATE width: d_model=%d n_head=%d d_ff=%d (built for 64/8/128)", d->d_model,
                    d->n_head, d->d_ff);
    if (d->n_layers < 1 || d->n_layers > 64) return fail(DNAG_E_ARG, "unsupported n_layers=%d", d->n_layers);
    if (!is_nb201(*d) && !is_ofa(*d))
        return fail(DNAG_E_ARG, "unsupported graph shape max_node=%d n_vocab=%d (built for 8x7 and 20x9)", d->max_node,
                    d->n_vocab);
    return 0;
}

static int check_pred(const dnag_pred_dims *d) {
    if (!d) return fail(DNAG_E_ARG, "pred dims is NULL");
    if (d->max_node != 8 || d->n_vocab != 7)
        return fail(DNAG_E_ARG, "MetaNeuralPredictor kernel is built for 8x7 graphs, got %dx%d", d->max_node, d->n_vocab);
    if (d->gcn_hidden != PRED_GH || d->linear_hidden != 128)
        return fail(DNAG_E_ARG, "unsupported surrogate width gcn_hidden=%d linear_hidden=%d (built for 144/128)",
                    d->gcn_hidden, d->linear_hidden);
    if (d->gcn_layers < 1 || d->gcn_layers > DNAG_MAX_GCN) return fail(DNAG_E_ARG, "unsupported gcn_layers=%d", d->gcn_layers);
    if (d->hs < 1 || d->hs > PRED_MAXHS) return fail(DNAG_E_ARG, "unsupported hs=%d (max %d)", d->hs, PRED_MAXHS);
    if (d->nz < 1 || d->nz > 1024) return fail(DNAG_E_ARG, "unsupported nz=%d", d->nz);
    return 0;
}

static int check_predcate(const dnag_predcate_dims *d) {
    if (!d) return fail(DNAG_E_ARG, "predcate dims is NULL");
    if (d->max_node != 20 || d->n_vocab != 9)
        return fail(DNAG_E_ARG, "MetaPredictorCATE kernels are built for 20x9 graphs, got %dx%d", d->max_node, d->n_vocab);
    if (d->d_model != 64 || d->n_head != 2 || d->d_ff != 32)
        return fail(DNAG_E_ARG, "unsupported MetaPredictorCATE width: d_model=%d n_head=%d d_ff=%d (built for 64/2/32)",
                    d->d_model, d->n_head, d->d_ff);
    if (d->n_layers < 1 || d->n_layers > DNAG_MAX_PCL) return fail(DNAG_E_ARG, "unsupported n_layers=%d", d->n_layers);
    if (d->hs < GEMM_BN || d->hs % GEMM_BN || d->hs > 4096) return fail(DNAG_E_ARG, "unsupported hs=%d (a multiple of %d)", d->hs, GEMM_BN);
    if (d->nz < 1 || d->nz > 1024) return fail(DNAG_E_ARG, "unsupported nz=%d", d->nz);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// packing (host only)
// ---------------------------------------------------------------------------------------------
extern "C" size_t dnag_cate_state_floats(const dnag_cate_dims *c) {
    if (!c) return 0;
    const size_t d = c->d_model, V = c->n_vocab, dff = c->d_ff, L = c->n_layers, F = L * d, Hd = 2 * F;
    size_t n = d * V + d + 4 * d * d + 4 * d + 4 * d * d + d;
    n += L * (2 * d + 4 * (d * d + d) + 2 * d + dff * d + dff + d * dff + d);
    n += L * 2 * d;
    n += Hd * F + Hd + Hd * Hd + Hd + V * Hd + V;
    return n;
}
extern "C" size_t dnag_cate_table_floats(const dnag_cate_dims *c) {
    return c ? (size_t)c->max_node * c->d_model + c->d_model / 2 : 0;
}
extern "C" size_t dnag_cate_packed_floats(const dnag_cate_dims *c) { return c ? cate_layout(*c).total : 0; }

// fold gamma/beta of a LayerNorm into the linear that consumes it: W' = W diag(gamma), b' = b + W beta
static void fold_ln(const float *W, const float *b, const float *gam, const float *bet, size_t rows, size_t d, float *Wo, float *bo) {
    for (size_t r = 0; r < rows; ++r) {
        double acc = b[r];
        for (size_t k = 0; k < d; ++k) {
            Wo[r * d + k] = W[r * d + k] * gam[k];
            acc += (double)W[r * d + k] * (double)bet[k];
        }
        if (bo) bo[r] = (float)acc;
    }
}

static float amax_of(const float *w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(w[i]));
    return m;
}
// worst-case |y_c| of y = W u + b over LayerNorm outputs u (no affine: ||u||_2 <= sqrt(d), dnag_common.h)
static float bound_ln_linear(const float *W, const float *b, size_t rows, size_t d) {
    double m = 0.0;
    for (size_t r = 0; r < rows; ++r) {
        double q = 0.0;
        for (size_t k = 0; k < d; ++k) q += (double)W[r * d + k] * (double)W[r * d + k];
        m = fmax(m, sqrt((double)d) * sqrt(q) + fabs((double)b[r]));
    }
    return (float)(m * 1.0001);
}
// worst-case |y_c| of y = W a + b over inputs with |a_k| <= amax
static float bound_l1_linear(const float *W, const float *b, size_t rows, size_t K, float amax) {
    double m = 0.0;
    for (size_t r = 0; r < rows; ++r) {
        double q = 0.0;
        for (size_t k = 0; k < K; ++k) q += fabs((double)W[r * K + k]);
        m = fmax(m, q * (double)amax + fabs((double)b[r]));
    }
    return (float)(m * 1.0001);
}

// state_dict order -> CateLayout; `pe` may be NULL (no positional table: zeros)
static int cate_pack_impl(const dnag_cate_dims &c, const float *s, const float *pe, const float *freq, float *out) {
    const CateLayout y = cate_layout(c);
    const size_t d = y.d, V = y.V, dff = y.dff, L = y.L, F = y.F, Hd = y.Hd;
    for (size_t i = 0, n = dnag_cate_state_floats(&c); i < n; ++i)
        if (!std::isfinite(s[i])) return fail(DNAG_E_ARG, "state_flat[%zu] = %g is not finite", i, (double)s[i]);
    memset(out, 0, y.total * sizeof(float));
    size_t o = 0;
    auto take = [&](size_t dst, size_t n) { memcpy(out + dst, s + o, n * sizeof(float)); o += n; };
    take(y.emb_w, d * V); take(y.emb_b, d);
    take(y.t1_w, 4 * d * d); take(y.t1_b, 4 * d);
    take(y.t2_w, 4 * d * d); take(y.t2_b, d);
    std::vector<float> tmp((3 * d > dff ? 3 * d : dff) * d);
    // Power-of-two operand scales of the split arithmetic, from the weights alone (dnag_common.h, "Operand scales")
    for (size_t l = 0; l < L; ++l) {
        float *LP = out + y.layer0 + l * y.layer_stride;
        float *SC = LP + y.l_sc;
        const float *ln1g = s + o, *ln1b = s + o + d; o += 2 * d;
        for (int j = 0; j < 3; ++j) {   // Q, K, V
            fold_ln(s + o, s + o + d * d, ln1g, ln1b, d, d, tmp.data() + j * d * d, LP + y.l_bqkv + j * d);
            o += d * d + d;
        }
        const float sw_qkv = weight_scale(amax_of(tmp.data(), 3 * d * d));
        const float sa_o = act_scale(bound_ln_linear(tmp.data() + 2 * d * d, LP + y.l_bqkv + 2 * d, d, d));   // attention output: convex combinations of V rows
        frag_pack(tmp.data(), 3 * d, d, d, LP + y.l_wqkv);
        frag16_pack(tmp.data(), 3 * d, d, d, LP + y.delta16 + y.l_wqkv, sw_qkv);
        const float sw_o = weight_scale(amax_of(s + o, d * d));
        frag_pack(s + o, d, d, d, LP + y.l_wo);
        frag16_pack(s + o, d, d, d, LP + y.delta16 + y.l_wo, sw_o); o += d * d;
        memcpy(LP + y.l_bo, s + o, d * sizeof(float)); o += d;
        const float *ln2g = s + o, *ln2b = s + o + d; o += 2 * d;
        fold_ln(s + o, s + o + dff * d, ln2g, ln2b, dff, d, tmp.data(), LP + y.l_b1); o += dff * d + dff;
        const float sw_1 = weight_scale(amax_of(tmp.data(), dff * d));
        const float sa_f2 = act_scale(bound_ln_linear(tmp.data(), LP + y.l_b1, dff, d));   // relu does not raise the bound
        frag_pack(tmp.data(), dff, d, d, LP + y.l_w1);
        frag16_pack(tmp.data(), dff, d, d, LP + y.delta16 + y.l_w1, sw_1);
        const float sw_2 = weight_scale(amax_of(s + o, d * dff));
        frag_pack(s + o, d, dff, dff, LP + y.l_w2);
        frag16_pack(s + o, d, dff, dff, LP + y.delta16 + y.l_w2, sw_2); o += d * dff;
        memcpy(LP + y.l_b2, s + o, d * sizeof(float)); o += d;
        SC[ENC_SC_INV_QKV] = 1.0f / (16.0f * sw_qkv);
        SC[ENC_SC_SA_O] = sa_o;
        SC[ENC_SC_INV_O] = 1.0f / (sa_o * sw_o);
        SC[ENC_SC_INV_F1] = 1.0f / (16.0f * sw_1);
        SC[ENC_SC_SA_F2] = sa_f2;
        SC[ENC_SC_INV_F2] = 1.0f / (sa_f2 * sw_2);
    }
    for (size_t l = 0; l < L; ++l) {   // encoder_f.norms.l
        float *LP = out + y.layer0 + l * y.layer_stride;
        memcpy(LP + y.l_ong, s + o, d * sizeof(float)); o += d;
        memcpy(LP + y.l_onb, s + o, d * sizeof(float)); o += d;
        float *vec = LP + y.l_vec;     // contiguous copy of the layer's vectors
        memcpy(vec, LP + y.l_bqkv, 3 * d * sizeof(float)); vec += 3 * d;
        memcpy(vec, LP + y.l_bo, d * sizeof(float)); vec += d;
        memcpy(vec, LP + y.l_b1, dff * sizeof(float)); vec += align4(dff);
        memcpy(vec, LP + y.l_b2, d * sizeof(float)); vec += d;
        memcpy(vec, LP + y.l_ong, d * sizeof(float)); vec += d;
        memcpy(vec, LP + y.l_onb, d * sizeof(float));
    }
    take(y.f0_w, Hd * F); take(y.f0_b, Hd);
    take(y.f1_w, Hd * Hd); take(y.f1_b, Hd);
    take(y.f2_w, V * Hd); take(y.f2_b, V);
    {
        // final.linears.0 on the concat of the per-layer output norms: o_l = LN(h) gamma_l + beta_l.  The affine is folded
        // into the split copy (W0' = W0 diag(gamma), b0' = b0 + W0 beta), so the encoders write LN(h) itself.
        std::vector<float> w0f(Hd * F);
        for (size_t j = 0; j < Hd; ++j) {
            double acc = out[y.f0_b + j];
            for (size_t l = 0; l < L; ++l) {
                const float *og = out + y.layer0 + l * y.layer_stride + y.l_ong, *ob = out + y.layer0 + l * y.layer_stride + y.l_onb;
                for (size_t k = 0; k < d; ++k) {
                    const float w = out[y.f0_w + j * F + l * d + k];
                    w0f[j * F + l * d + k] = w * og[k];
                    acc += (double)w * (double)ob[k];
                }
            }
            out[y.f0_bs + j] = (float)acc;
        }
        float *H = out + y.hsc;
        const float cbound = sqrtf((float)d);          // |LN(h)| < sqrt(d) = 8
        const float sa_c = 16.0f;                       // the scale of every LayerNorm image (act_scale(8) = 16 as well)
        const float sw0 = weight_scale(amax_of(w0f.data(), Hd * F)), sw1 = weight_scale(amax_of(out + y.f1_w, Hd * Hd));
        const float sw3 = weight_scale(amax_of(out + y.f2_w, V * Hd));
        // |elu(z)| <= max(|z|, 1)
        const float h1b = fmaxf(1.0f, bound_l1_linear(w0f.data(), out + y.f0_bs, Hd, F, cbound));
        const float h2b = fmaxf(1.0f, bound_l1_linear(out + y.f1_w, out + y.f1_b, Hd, Hd, h1b));
        const float sa1 = act_scale(h1b), sa2 = act_scale(h2b);
        H[HEAD_SC_SA_CONCAT] = sa_c;
        H[HEAD_SC_F0 + 0] = 1.0f / (sa_c * sw0); H[HEAD_SC_F0 + 1] = sa1;
        H[HEAD_SC_F1 + 0] = 1.0f / (sa1 * sw1); H[HEAD_SC_F1 + 1] = sa2; H[HEAD_SC_F1 + 2] = sw3; H[HEAD_SC_F1 + 3] = 1.0f / (sa2 * sw3);
        split_pack(w0f.data(), Hd, F, out + y.f0_s, sw0);
        split_pack(out + y.f1_w, Hd, Hd, out + y.f1_s, sw1);
        kblock_rows(out + y.f0_s, Hd, F);
        kblock_rows(out + y.f1_s, Hd, Hd);
    }
    if (o != dnag_cate_state_floats(&c)) return fail(DNAG_E_ARG, "internal: consumed %zu of %zu floats", o, dnag_cate_state_floats(&c));
    if (pe) memcpy(out + y.pe, pe, (size_t)y.N * d * sizeof(float));
    memcpy(out + y.freq, freq, (d / 2) * sizeof(float));
    return 0;
}

extern "C" int dnag_cate_pack(const dnag_cate_dims *c, const float *s, const float *tables, float *out) {
    if (int rc = check_cate(c)) return rc;
    if (!s || !tables || !out) return fail(DNAG_E_ARG, "NULL pointer");
    return cate_pack_impl(*c, s, tables, tables + (size_t)c->max_node * c->d_model, out);
}

extern "C" size_t dnag_pred_state_floats(const dnag_pred_dims *c) {
    if (!c) return 0;
    const size_t g = c->gcn_hidden, V = c->n_vocab, L = c->gcn_layers, lh = c->linear_hidden, hs = c->hs, nz = c->nz;
    return 2 * V * g + (L - 1) * 2 * g * g + lh * g + 4 * g * g + 4 * g + 4 * g * g + g + hs * (nz + lh) + hs + hs + 1;
}
extern "C" size_t dnag_pred_table_floats(const dnag_pred_dims *c) { return c ? (size_t)c->gcn_hidden / 2 : 0; }
extern "C" size_t dnag_pred_packed_floats(const dnag_pred_dims *c) { return c ? pred_layout(*c).total : 0; }

extern "C" int dnag_pred_pack(const dnag_pred_dims *c, const float *s, const float *tables, float *out) {
    if (int rc = check_pred(c)) return rc;
    if (!s || !tables || !out) return fail(DNAG_E_ARG, "NULL pointer");
    const PredLayout y = pred_layout(*c);
    const size_t g = y.g, lh = y.lh, hs = y.hs, nz = y.nz;
    memset(out, 0, y.total * sizeof(float));
    size_t o = 0;
    for (size_t i = 0, n = dnag_pred_state_floats(c); i < n; ++i)
        if (!std::isfinite(s[i])) return fail(DNAG_E_ARG, "state_flat[%zu] = %g is not finite", i, (double)s[i]);
    std::vector<float> tr(g * (g > lh ? g : lh));
    // every split-arithmetic matrix gets its own power-of-two scale (dnag_common.h "Operand scales"); the kernel divides
    // it out through the table at y.wsc.  The activations' scales are derived at run time (predictor_kernel, phase_scale).
    for (int l = 0; l < y.L; ++l) {
        const size_t in = l == 0 ? (size_t)y.V : g;
        for (int w = 0; w < 2; ++w) {
            const float *W = s + o; o += in * g;   // [in][g]
            const float sw = weight_scale(amax_of(W, in * g));
            out[y.wsc + 2 * l + w] = 1.0f / sw;
            {   // bounds of the GEMM accumulators that feed the adjacency mixing (PredLayout::mixs)
                double col1 = 0.0, row1 = 0.0;
                for (size_t k = 0; k < g; ++k) { double a = 0.0; for (size_t i = 0; i < in; ++i) a += fabs((double)W[i * g + k]); col1 = fmax(col1, a); }
                for (size_t i = 0; i < in; ++i) { double a = 0.0; for (size_t k = 0; k < g; ++k) a += fabs((double)W[i * g + k]); row1 = fmax(row1, a); }
                float *ms = out + y.mixs + (2 * l + w) * 4;
                const float sf = pow2_scale_of((float)(1024.0 * sw * col1 * 1.0001)), sb = pow2_scale_of((float)(1024.0 * sw * row1 * 1.0001));
                ms[0] = sf; ms[1] = 1.0f / (sf * 1024.0f); ms[2] = sb; ms[3] = 1.0f / (sb * 1024.0f);
            }
            if (l == 0) {
                memcpy(out + (w ? y.w2_0 : y.w1_0), W, in * g * sizeof(float));
                std::vector<float> t16(g * 16, 0.f);            // W^T [g][V -> 16]: layer 0 forward, K = 16
                for (size_t i = 0; i < in; ++i)
                    for (size_t k = 0; k < g; ++k) t16[k * 16 + i] = W[i * g + k];
                frag16_pack(t16.data(), g, 16, 16, out + (w ? y.w2f0 : y.w1f0), sw);
            }
            frag_pack(W, in, g, g, out + (w ? y.w2b[l] : y.w1b[l]));
            frag16_pack(W, in, g, g, out + y.delta16 + (w ? y.w2b[l] : y.w1b[l]), sw);
            if (l > 0) {
                for (size_t i = 0; i < in; ++i)
                    for (size_t k = 0; k < g; ++k) tr[k * in + i] = W[i * g + k];
                frag_pack(tr.data(), g, in, in, out + (w ? y.w2f[l] : y.w1f[l]));
                frag16_pack(tr.data(), g, in, in, out + y.delta16 + (w ? y.w2f[l] : y.w1f[l]), sw);
            }
        }
    }
    {
        const float *W = s + o; o += lh * g;   // fc1 [lh][g]
        const float sw = weight_scale(amax_of(W, lh * g));
        out[y.wsc + 2 * y.L] = 1.0f / sw;
        frag_pack(W, lh, g, g, out + y.fc1f);
        frag16_pack(W, lh, g, g, out + y.delta16 + y.fc1f, sw);
        for (size_t i = 0; i < lh; ++i)
            for (size_t k = 0; k < g; ++k) tr[k * lh + i] = W[i * g + k];
        frag_pack(tr.data(), g, lh, lh, out + y.fc1b);
        frag16_pack(tr.data(), g, lh, lh, out + y.delta16 + y.fc1b, sw);
    }
    memcpy(out + y.t1_w, s + o, 4 * g * g * sizeof(float)); o += 4 * g * g;
    memcpy(out + y.t1_b, s + o, 4 * g * sizeof(float)); o += 4 * g;
    memcpy(out + y.t2_w, s + o, 4 * g * g * sizeof(float)); o += 4 * g * g;
    memcpy(out + y.t2_b, s + o, g * sizeof(float)); o += g;
    {
        const float *W = s + o; o += hs * (nz + lh);   // pred_fc.0 [hs][nz+lh]
        memcpy(out + y.p0_w, W, hs * (nz + lh) * sizeof(float));
        for (size_t j = 0; j < hs; ++j)
            for (size_t a = 0; a < lh; ++a) out[y.p0_at + a * hs + j] = W[j * (nz + lh) + nz + a];
        // the arch half W0a [hs][lh] and its transpose [lh][hs16] as fp16 hi/lo fragments
        const size_t hs16 = (size_t)y.hs16;
        std::vector<float> wa(hs * lh), wt(lh * hs16, 0.f);
        for (size_t j = 0; j < hs; ++j)
            for (size_t a = 0; a < lh; ++a) { wa[j * lh + a] = W[j * (nz + lh) + nz + a]; wt[a * hs16 + j] = wa[j * lh + a]; }
        const float sw = weight_scale(amax_of(wa.data(), hs * lh));
        out[y.wsc + 2 * y.L + 1] = 1.0f / sw;
        frag16_pack(wa.data(), hs, lh, lh, out + y.p0f, sw);
        frag16_pack(wt.data(), lh, hs16, hs16, out + y.p0b, sw);
    }
    memcpy(out + y.p0_b, s + o, hs * sizeof(float)); o += hs;
    memcpy(out + y.p2_w, s + o, hs * sizeof(float)); o += hs;
    out[y.p2_b] = s[o]; o += 1;
    if (o != dnag_pred_state_floats(c)) return fail(DNAG_E_ARG, "internal: consumed %zu of %zu floats", o, dnag_pred_state_floats(c));
    memcpy(out + y.freq, tables, (g / 2) * sizeof(float));
    return 0;
}

extern "C" size_t dnag_predcate_state_floats(const dnag_predcate_dims *c) {
    if (!c) return 0;
    const dnag_cate_dims e = predcate_as_cate(*c);
    const size_t R = (size_t)c->max_node * c->n_vocab, hs = c->hs, nz = c->nz;
    return dnag_cate_state_floats(&e) + (2 * R * R + 2 * R) + (2 * R * 2 * R + 2 * R) + hs * (nz + 2 * R) + hs + hs + 1;
}
extern "C" size_t dnag_predcate_table_floats(const dnag_predcate_dims *c) { return c ? (size_t)c->d_model / 2 : 0; }
extern "C" size_t dnag_predcate_packed_floats(const dnag_predcate_dims *c) { return c ? predcate_layout(*c).total : 0; }

extern "C" int dnag_predcate_pack(const dnag_predcate_dims *c, const float *s, const float *tables, float *out) {
    if (int rc = check_predcate(c)) return rc;
    if (!s || !tables || !out) return fail(DNAG_E_ARG, "NULL pointer");
    const PredCateLayout y = predcate_layout(*c);
    const dnag_cate_dims e = predcate_as_cate(*c);
    memset(out, 0, y.total * sizeof(float));
    if (int rc = cate_pack_impl(e, s, nullptr, tables, out)) return rc;
    const size_t d = y.enc.d, dff = y.enc.dff, L = y.enc.L, F = y.enc.F, Hd = y.enc.Hd, V = y.enc.V;
    const size_t R = y.R, RK = y.RK, R2 = y.R2, R2P = y.R2P, hs = y.hs, nz = y.nz;
    // transposed operands of the encoder's backward pass (second walk over the layer parameters)
    size_t o = d * V + d + 4 * d * d + 4 * d + 4 * d * d + d;
    std::vector<float> w((3 * d > dff ? 3 * d : dff) * d), t(3 * d * d);
    auto transpose = [&](const float *W, size_t rows, size_t cols) {   // W [rows][cols] -> t [cols][rows]
        for (size_t i = 0; i < rows; ++i)
            for (size_t k = 0; k < cols; ++k) t[k * rows + i] = W[i * cols + k];
    };
    for (size_t l = 0; l < L; ++l) {
        float *LB = out + y.bl[l];
        const float *ln1g = s + o, *ln1b = s + o + d; o += 2 * d;
        for (int j = 0; j < 3; ++j) { fold_ln(s + o, s + o + d * d, ln1g, ln1b, d, d, w.data() + j * d * d, nullptr); o += d * d + d; }
        transpose(w.data(), 3 * d, d);                      // Wqkv' [3d][d] -> [d][3d]
        frag_pack(t.data(), d, 3 * d, 3 * d, LB + y.b_wqkvT);
        transpose(s + o, d, d); o += d * d + d;             // Wo [d][d]
        frag_pack(t.data(), d, d, d, LB + y.b_woT);
        const float *ln2g = s + o, *ln2b = s + o + d; o += 2 * d;
        fold_ln(s + o, s + o + dff * d, ln2g, ln2b, dff, d, w.data(), nullptr); o += dff * d + dff;
        transpose(w.data(), dff, d);                        // W1' [dff][d] -> [d][dff]
        frag_pack(t.data(), d, dff, dff, LB + y.b_w1T);
        transpose(s + o, d, dff); o += d * dff + d;         // W2 [d][dff] -> [dff][d]
        frag_pack(t.data(), dff, d, d, LB + y.b_w2T);
    }
    o += L * 2 * d;
    for (size_t j = 0; j < Hd; ++j)                          // final.linears.0 [Hd][F] -> [F][Hd]
        for (size_t k = 0; k < F; ++k) out[y.f0T + k * Hd + j] = s[o + j * F + k];
    o += Hd * F + Hd;
    for (size_t j = 0; j < Hd; ++j)                          // final.linears.1 [Hd][Hd] -> transposed
        for (size_t k = 0; k < Hd; ++k) out[y.f1T + k * Hd + j] = s[o + j * Hd + k];
    o += Hd * Hd + Hd;
    o += V * Hd + V;
    if (o != dnag_cate_state_floats(&e)) return fail(DNAG_E_ARG, "internal: encoder walk consumed %zu of %zu floats", o, dnag_cate_state_floats(&e));
    for (size_t j = 0; j < R2; ++j)                          // regeress.linears.0 [2R][R]
        for (size_t k = 0; k < R; ++k) {
            out[y.r0_w + j * RK + k] = s[o + j * R + k];
            out[y.r0T + k * R2P + j] = s[o + j * R + k];
        }
    o += R2 * R;
    memcpy(out + y.r0_b, s + o, R2 * sizeof(float)); o += R2;
    for (size_t j = 0; j < R2; ++j)                          // regeress.linears.1 [2R][2R]
        for (size_t k = 0; k < R2; ++k) {
            out[y.r1_w + j * R2P + k] = s[o + j * R2 + k];
            out[y.r1T + k * R2P + j] = s[o + j * R2 + k];
        }
    o += R2 * R2;
    memcpy(out + y.r1_b, s + o, R2 * sizeof(float)); o += R2;
    for (size_t j = 0; j < hs; ++j) {                        // pred_fc.0 [hs][nz + 2R]
        const float *row = s + o + j * (nz + R2);
        memcpy(out + y.p0_wd + j * nz, row, nz * sizeof(float));
        for (size_t k = 0; k < R2; ++k) {
            out[y.p0_wa + j * R2P + k] = row[nz + k];
            out[y.p0T + k * hs + j] = row[nz + k];
        }
    }
    o += hs * (nz + R2);
    memcpy(out + y.p0_b, s + o, hs * sizeof(float)); o += hs;
    memcpy(out + y.p2_w, s + o, hs * sizeof(float)); o += hs;
    out[y.p2_b] = s[o]; o += 1;
    if (o != dnag_predcate_state_floats(c)) return fail(DNAG_E_ARG, "internal: consumed %zu of %zu floats", o, dnag_predcate_state_floats(c));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
struct WsLayout {
    size_t rows, rows_pad;
    size_t labels, temb_s, temb_p, c0, concat, h1, h2, cgrad, gbuf, x, x_mean, partials, pc, total;   // float offsets
    int nblk_exit;
};

// workspace of one MetaPredictorCATE evaluation (float offsets relative to its base)
struct PcWs {
    size_t rows, rows_pad, bpad;
    size_t beff, concat, h1, h2, flat, r1, r2, pz, dr2, dr1, dflat, dh1, dconcat, sv_h, sv_h1, sv_qkv, sv_f, amax, total;
};

static size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int exit_gf(const dnag_cate_dims &c) { return is_nb201(c) ? 8 : 3; }

static PcWs predcate_ws(const PredCateLayout &y, int batch) {
    PcWs w;
    const size_t N = y.enc.N, d = y.enc.d, F = y.enc.F, Hd = y.enc.Hd, L = y.enc.L;
    w.rows = (size_t)batch * N;
    w.rows_pad = up(w.rows, 256);        // the split-arithmetic GEMMs of the final MLP run 256-row tiles
    w.bpad = up((size_t)batch, GEMM_BM);
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += up(n, 64); return at; };
    w.amax = take(8);         // max |W0|, |W1|, |concat|, |H1|, |dH2| (float bits): operand scales of the split GEMMs
    w.beff = take(y.hs);
    w.concat = take(w.rows_pad * F);
    w.h1 = take(w.rows_pad * Hd);
    w.h2 = take(w.rows_pad * Hd);
    w.flat = take(w.bpad * y.RK);
    w.r1 = take(w.bpad * y.R2P);
    w.r2 = take(w.bpad * y.R2P);
    w.pz = take(w.bpad * y.hs);
    w.dr2 = take(w.bpad * y.R2P);
    w.dr1 = take(w.bpad * y.R2P);
    w.dflat = take(w.bpad * y.RK);
    w.dh1 = take(w.rows_pad * Hd);
    w.dconcat = take(w.rows_pad * F);
    w.sv_h = take((L + 1) * w.rows_pad * d);
    w.sv_h1 = take(L * w.rows_pad * d);
    w.sv_qkv = take(L * w.rows_pad * 3 * d);
    w.sv_f = take(L * w.rows_pad * y.enc.dff);
    w.total = o;
    return w;
}

static WsLayout ws_layout(const dnag_cate_dims &c, const dnag_pred_dims *p, int batch, int n_steps,
                          const dnag_predcate_dims *pc = nullptr) {
    WsLayout w;
    const size_t N = c.max_node, V = c.n_vocab, F = (size_t)c.n_layers * c.d_model, Hd = 2 * F;
    const size_t nt = (size_t)(n_steps > batch ? n_steps : batch);
    w.rows = (size_t)batch * N;
    w.rows_pad = up(w.rows, 256);   // the split GEMM runs 256-row tiles
    w.nblk_exit = (batch + exit_gf(c) - 1) / exit_gf(c);
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += up(n, 64); return at; };
    w.labels = take(nt);
    w.temb_s = take(nt * c.d_model);
    w.temb_p = take(p ? nt * p->gcn_hidden : pc ? nt * pc->d_model : 0);
    w.c0 = take(p ? (size_t)p->hs : 0);
    w.concat = take(w.rows_pad * F);
    w.h1 = take(w.rows_pad * Hd);
    w.h2 = take(w.rows_pad * Hd);
    w.cgrad = take((size_t)batch * N * V);
    w.gbuf = take((size_t)batch * N * V);
    w.x = take((size_t)batch * N * V);
    w.x_mean = take((size_t)batch * N * V);
    w.partials = take(2 * (size_t)w.nblk_exit);
    w.pc = o;
    if (pc) o += predcate_ws(predcate_layout(*pc), batch).total;
    w.total = o;
    return w;
}

extern "C" size_t dnag_langevin_partials_floats(const dnag_cate_dims *cd, int batch) {
    if (!cd || batch < 1) return 0;
    return 2 * (size_t)((batch + exit_gf(*cd) - 1) / exit_gf(*cd));
}
extern "C" size_t dnag_workspace_bytes(const dnag_cate_dims *cd, const dnag_pred_dims *pd, int batch, int n_steps) {
    if (!cd || batch < 1) return 0;
    return ws_layout(*cd, pd, batch, n_steps < 1 ? 1 : n_steps).total * sizeof(float);
}
extern "C" size_t dnag_workspace_bytes_predcate(const dnag_cate_dims *cd, const dnag_predcate_dims *pd, int batch, int n_steps) {
    if (!cd || batch < 1) return 0;
    return ws_layout(*cd, nullptr, batch, n_steps < 1 ? 1 : n_steps, pd).total * sizeof(float);
}
extern "C" size_t dnag_predcate_workspace_bytes(const dnag_predcate_dims *d, int batch) {
    if (!d || batch < 1) return 0;
    return (up((size_t)batch * d->d_model, 64) + predcate_ws(predcate_layout(*d), batch).total) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
// kernel timing (bench.py roofline): event pairs around every `every`-th head GEMM (K = 2F) launch
// ---------------------------------------------------------------------------------------------
struct Timing {
    bool on = false;
    int every = 1, max_samples = 0, seen = 0;
    std::vector<hipEvent_t> ev0, ev1;
    int used = 0;
    double flops = 0.0;
};
static Timing g_tm;
static std::mutex g_tm_mu;

extern "C" int dnag_timing_begin(int every, int max_samples) {
    std::lock_guard<std::mutex> lk(g_tm_mu);
    if (g_tm.on) return fail(DNAG_E_ARG, "timing already enabled");
    if (every < 1 || max_samples < 1 || max_samples > 65536) return fail(DNAG_E_ARG, "bad timing arguments");
    g_tm.ev0.resize(max_samples); g_tm.ev1.resize(max_samples);
    for (int i = 0; i < max_samples; ++i) { HIP_TRY(hipEventCreate(&g_tm.ev0[i])); HIP_TRY(hipEventCreate(&g_tm.ev1[i])); }
    g_tm.every = every; g_tm.max_samples = max_samples; g_tm.seen = 0; g_tm.used = 0; g_tm.flops = 0.0; g_tm.on = true;
    return 0;
}
extern "C" int dnag_timing_read(int *n_samples, double *mean_ms, double *flops_per_launch) {
    std::lock_guard<std::mutex> lk(g_tm_mu);
    if (!g_tm.on) return fail(DNAG_E_ARG, "timing not enabled");
    double tot = 0.0;
    for (int i = 0; i < g_tm.used; ++i) {
        HIP_TRY(hipEventSynchronize(g_tm.ev1[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_tm.ev0[i], g_tm.ev1[i]));
        tot += ms;
    }
    if (n_samples) *n_samples = g_tm.used;
    if (mean_ms) *mean_ms = g_tm.used ? tot / g_tm.used : 0.0;
    if (flops_per_launch) *flops_per_launch = g_tm.flops;
    return 0;
}
extern "C" int dnag_timing_end(void) {
    std::lock_guard<std::mutex> lk(g_tm_mu);
    if (!g_tm.on) return 0;
    for (int i = 0; i < g_tm.max_samples; ++i) { hipEventDestroy(g_tm.ev0[i]); hipEventDestroy(g_tm.ev1[i]); }
    g_tm.ev0.clear(); g_tm.ev1.clear(); g_tm.on = false;
    return 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device setting: set it once per (kernel, device), from whichever
// thread launches first there (two threads racing set the same value twice)
struct LdsAttr {
    std::atomic<size_t> bytes[64];
    LdsAttr() { for (auto &b : bytes) b.store(0); }
};
static int ensure_lds(LdsAttr &a, const void *kern, size_t bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::atomic<size_t> &have = a.bytes[dev & 63];
    if (have.load(std::memory_order_acquire) >= bytes) return 0;
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have.store(bytes, std::memory_order_release);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// C = act(A W^T + bias); `timed` marks the launches bench.py's roofline is computed on (final.linears.1)
template <int ACT, int TAG>
static int launch_gemm_t(const float *A, const float *W, const float *bias, float *C, const float *aux, size_t M, int Nout, int K,
                         int lda, int ldc, int nvalid, hipStream_t st, int splitk = 1) {
    if (M % GEMM_BM || Nout % GEMM_BN || K % (GEMM_BK * splitk) || lda % 4 || nvalid > Nout || nvalid > ldc)
        return fail(DNAG_E_ARG, "gemm shape %zu x %d x %d (lda %d ldc %d nvalid %d split %d) not tile aligned", M, Nout, K, lda, ldc, nvalid, splitk);
    if (splitk > 1 && (ACT != ACT_NONE || bias)) return fail(DNAG_E_ARG, "split-K gemm takes no bias / activation");
    const int grid = (int)(M / GEMM_BM) * (Nout / GEMM_BN);
    hipLaunchKernelGGL((gemm_bias_act_kernel<ACT, TAG>), dim3(grid, splitk), dim3(256), 0, st, A, W, bias, C, aux, (int)M, Nout, K, lda, ldc, nvalid);
    LAUNCH_CHECK();
    return 0;
}

static int launch_gemm(const float *A, const float *W, const float *bias, float *C, size_t M, int Nout, int K, bool timed,
                       size_t real_rows, hipStream_t st, const float *sc, const float *W3 = nullptr, int V = 0) {
    // the event handles are copied out under the lock: dnag_timing_end may clear the vectors from another thread (an
    // event destroyed between here and its record only fails that record)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) {
        std::lock_guard<std::mutex> lk(g_tm_mu);
        if (g_tm.on) {
            if (g_tm.seen % g_tm.every == 0 && g_tm.used < g_tm.max_samples) {
                ev0 = g_tm.ev0[g_tm.used];
                ev1 = g_tm.ev1[g_tm.used];
                g_tm.used++;
            }
            g_tm.seen++;
            g_tm.flops = 2.0 * (double)real_rows * (double)Nout * (double)K;
        }
    }
    if (ev0) HIP_TRY(hipEventRecord(ev0, st));
    // split-arithmetic fp16 MFMA GEMMs (dnag_gemm_dma.hip.h): final.linears.0 (K = F) reads the encoder's split-pair
    // concat and writes split-pair rows, both k-tile blocked; final.linears.1 (K = 2F, the timed one) reads those and
    // applies final.linears.2 in its epilogue.  Both weight matrices are k-tile blocked (kblock_rows at pack time).
    typedef DmaGeom Ge;
    if (M % Ge::BM || Nout % Ge::BN || K % Ge::BK) return fail(DNAG_E_ARG, "gemm shape %zu x %d x %d not tile aligned", M, Nout, K);
    const int grid = (int)(M / Ge::BM) * (Nout / Ge::BN);
    auto k7 = gemm_dma_kernel<ACT_ELU, false, 7, true>;      // final.linears.1 + final.linears.2 fused, NAS-Bench-201
    auto k9 = gemm_dma_kernel<ACT_ELU, false, 9, true>;      // same, OFA
    auto k0 = gemm_dma_kernel<ACT_ELU, true, 0, true>;
    static LdsAttr a7, a9, a0;
    if (int rc = ensure_lds(a7, reinterpret_cast<const void *>(k7), Ge::lds_bytes)) return rc;
    if (int rc = ensure_lds(a9, reinterpret_cast<const void *>(k9), Ge::lds_bytes)) return rc;
    if (int rc = ensure_lds(a0, reinterpret_cast<const void *>(k0), Ge::lds_bytes)) return rc;
    if (timed) {
        if (V != 7 && V != 9) return fail(DNAG_E_ARG, "fused head GEMM is built for 7 or 9 output columns, got %d", V);
        // C receives the partial sums [Nout / 256][M][V] of final.linears.2
        hipLaunchKernelGGL(V == 7 ? k7 : k9, dim3(grid), dim3(Ge::T), Ge::lds_bytes, st, A, W, bias, (float *)nullptr, (int)M, Nout, K, W3, C, sc);
    } else {
        hipLaunchKernelGGL(k0, dim3(grid), dim3(Ge::T), Ge::lds_bytes, st, A, W, bias, C, (int)M, Nout, K, (const float *)nullptr, (float *)nullptr, sc);
    }
    LAUNCH_CHECK();
    if (ev1) HIP_TRY(hipEventRecord(ev1, st));
    return 0;
}

// C [M][Nout] (pitch ldc) = act(A W^T + bias) on the split-arithmetic GEMM with fp32 operands split on the fly
// (dnag_gemm16.hip.h, IN_F32).  amax_a / amax_w: device max|x| of the operand (null: the fixed activation / weight scale).
template <int ACT, int TAG>
static int launch_gemm32(const float *A, int lda, const float *amax_a, const float *W, int ldw, const float *amax_w, const float *bias,
                         float *C, int ldc, const float *aux, size_t M, int Nout, int K, int splitk, hipStream_t st, float *amax_out = nullptr) {
    typedef SplitGeom Ge;
    if (M % Ge::BM || Nout % Ge::BN || K % (GEMM_BK * splitk) || lda % 4 || ldw % 4)
        return fail(DNAG_E_ARG, "split gemm shape %zu x %d x %d (lda %d ldw %d split %d) not tile aligned", M, Nout, K, lda, ldw, splitk);
    if (splitk > 1 && (ACT != ACT_NONE || bias)) return fail(DNAG_E_ARG, "split-K gemm takes no bias / activation");
    auto kern = gemm_split_kernel<ACT, false, TAG, 0, true>;
    static LdsAttr attr;
    if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), Ge::lds_bytes)) return rc;
    SplitF32Args fa;
    fa.sc = nullptr;
    fa.lda = lda; fa.ldw = ldw; fa.ldc = ldc; fa.amax_a = amax_a; fa.amax_w = amax_w; fa.aux = aux; fa.amax_out = amax_out;
    const int grid = (int)(M / Ge::BM) * (Nout / Ge::BN);
    hipLaunchKernelGGL(kern, dim3(grid, splitk), dim3(Ge::T), Ge::lds_bytes, st, A, W, bias, C, (int)M, Nout, K, (const float *)nullptr,
                       (float *)nullptr, fa);
    LAUNCH_CHECK();
    return 0;
}

static int launch_time_embed(const float *labels, int n, int d, const float *P, size_t freq, size_t w1, size_t b1, size_t w2,
                             size_t b2, float *out, hipStream_t st) {
    hipLaunchKernelGGL(time_embed_kernel, dim3(n), dim3(256), (size_t)5 * d * sizeof(float), st, labels, d, P + freq, P + w1,
                       P + b1, P + w2, P + b2, out);
    LAUNCH_CHECK();
    return 0;
}

template <int N, int V, int G>
static int launch_encoder_t(const CateLayout &lay, const float *P, const float *x, const float *temb, int tstride,
                            const float *mask, int mstride, float *concat, float *attn0, int batch, hipStream_t st) {
    typedef EncGeom<N, G, 128> Ge;
    const size_t lds = Ge::lds_floats * sizeof(float);
    auto kern = encoder_kernel<N, V, G, 8, 128, true, false, false, true, true>;   // split-arithmetic linears, concat in split-pair rows
    static LdsAttr attr;
    if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), lds)) return rc;
    const int grid = (batch + G - 1) / G;
    EncSave none;
    memset(&none, 0, sizeof(none));
    DropMasks nodrop;
    memset(&nodrop, 0, sizeof(nodrop));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, P, lay, x, temb, tstride, mask, mstride, concat, attn0, batch, none, nodrop);
    LAUNCH_CHECK();
    return 0;
}

template <int N, int V>
static int launch_encoder8(const CateLayout &lay, const float *P, const float *x, const float *temb, int tstride, const float *mask,
                           int mstride, float *concat, int batch, hipStream_t st) {
    typedef Enc8Geom<N> Ge;
    if (lay.d != Ge::D || lay.dff != Ge::DFF) return fail(DNAG_E_ARG, "encoder8: d_model %d / d_ff %d", lay.d, lay.dff);
    const size_t lds = Ge::lds_floats * sizeof(float);
    auto kern = encoder8_kernel<N, V>;
    static LdsAttr attr;
    if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), lds)) return rc;
    EncSave nosave;
    DropMasks nodrop;
    memset(&nosave, 0, sizeof(nosave));
    memset(&nodrop, 0, sizeof(nodrop));
    hipLaunchKernelGGL(kern, dim3((batch + Ge::G - 1) / Ge::G), dim3(256), lds, st, P, lay, x, temb, tstride, mask, mstride, concat, batch, nosave, nodrop);
    LAUNCH_CHECK();
    return 0;
}

static int launch_encoder(const dnag_cate_dims &c, const CateLayout &lay, const float *P, const float *x, const float *temb,
                          int tstride, const float *mask, int mstride, float *concat, float *attn0, int batch, hipStream_t st) {
    if (attn0 == nullptr) {       // the sampling path: dnag_encoder8.hip.h
        if (is_nb201(c)) return launch_encoder8<8, 7>(lay, P, x, temb, tstride, mask, mstride, concat, batch, st);
        return launch_encoder8<20, 9>(lay, P, x, temb, tstride, mask, mstride, concat, batch, st);
    }
    if (is_nb201(c)) return launch_encoder_t<8, 7, 8>(lay, P, x, temb, tstride, mask, mstride, concat, attn0, batch, st);
    return launch_encoder_t<20, 9, 3>(lay, P, x, temb, tstride, mask, mstride, concat, attn0, batch, st);
}

static int launch_exit(const dnag_cate_dims &c, const CateLayout &lay, const float *P, const float *h2, const ExitArgs &a,
                       hipStream_t st) {
    const int gf = exit_gf(c);
    const int grid = (a.batch + gf - 1) / gf;
    if (is_nb201(c))
        hipLaunchKernelGGL((head_out_kernel<8, 7, 8>), dim3(grid), dim3(256), 0, st, h2, P + lay.f2_w, P + lay.f2_b, lay.Hd, a);
    else
        hipLaunchKernelGGL((head_out_kernel<20, 9, 3>), dim3(grid), dim3(256), 0, st, h2, P + lay.f2_w, P + lay.f2_b, lay.Hd, a);
    LAUNCH_CHECK();
    return 0;
}

// c0 [hs] = the dataset half of pred_fc.0, once per call (the set-encoder's D_mu is constant over a sampling call)
static int launch_pred_c0(const PredLayout &lay, const float *P, const float *d_mu, float *c0, hipStream_t st) {
    hipLaunchKernelGGL(pred_c0_kernel, dim3(lay.hs), dim3(64), 0, st, P, lay, d_mu, c0);
    LAUNCH_CHECK();
    return 0;
}

static int launch_predictor(const PredLayout &lay, const float *P, const float *x, const float *temb, int tstride,
                            const float *mask, int mstride, const float *c0, float *y, float *grad, int batch, hipStream_t st) {
    typedef PredGeom<8, 7> Ge;
    const size_t lds = Ge::lds_bytes(lay.hs, lay.L);
    auto kern = predictor_kernel<8, 7>;
    static LdsAttr attr;
    if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), lds)) return rc;
    const int grid = (batch + Ge::G - 1) / Ge::G;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PRED_NW * 64), lds, st, P, lay, x, temb, tstride, mask, mstride, c0, y, grad, batch);
    LAUNCH_CHECK();
    return 0;
}

// score-network body: x -> H2 (everything but the exit kernel)
static int run_score_body(const dnag_cate_dims &c, const CateLayout &lay, const float *P, const float *x, const float *temb,
                          int tstride, const float *mask, int mstride, float *ws, const WsLayout &w, float *attn0, int batch,
                          hipStream_t st) {
    if (int rc = launch_encoder(c, lay, P, x, temb, tstride, mask, mstride, ws + w.concat, attn0, batch, st)) return rc;
    if (int rc = launch_gemm(ws + w.concat, P + lay.f0_s, P + lay.f0_bs, ws + w.h1, w.rows_pad, lay.Hd, lay.F, false, w.rows, st, P + lay.hsc + HEAD_SC_F0)) return rc;
    // final.linears.1 with final.linears.2 fused into its epilogue: ws.h2 receives [Hd / 256][rows_pad][V] partial sums
    if (int rc = launch_gemm(ws + w.h1, P + lay.f1_s, P + lay.f1_b, ws + w.h2, w.rows_pad, lay.Hd, lay.Hd, true, w.rows, st, P + lay.hsc + HEAD_SC_F1, P + lay.f2_w, lay.V)) return rc;
    return 0;
}

extern "C" int dnag_cate_forward(const dnag_cate_dims *d, const float *packed, const float *x, const float *time_cond,
                                 int time_stride, const float *mask, int mask_stride, float *out, float *concat_out,
                                 float *attn0_out, int batch, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_cate(d)) return rc;
    if (!packed || !x || !time_cond || !mask || !out || !workspace) return fail(DNAG_E_ARG, "NULL pointer");
    if (batch < 1) return fail(DNAG_E_ARG, "batch=%d", batch);
    const CateLayout lay = cate_layout(*d);
    const WsLayout w = ws_layout(*d, nullptr, batch, 1);
    if (workspace_bytes < w.total * sizeof(float))
        return fail(DNAG_E_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, w.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const int nt = time_stride ? batch : 1;
    if (int rc = launch_time_embed(time_cond, nt, lay.d, packed, lay.freq, lay.t1_w, lay.t1_b, lay.t2_w, lay.t2_b, ws + w.temb_s, st)) return rc;
    if (int rc = run_score_body(*d, lay, packed, x, ws + w.temb_s, time_stride ? 1 : 0, mask, mask_stride ? 1 : 0, ws, w, attn0_out, batch, st)) return rc;
    ExitArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = EXIT_FORWARD; a.batch = batch; a.out = out;
    a.parts = ws + w.h2; a.nparts = lay.Hd / SplitGeom::BN; a.part_stride = w.rows_pad * (size_t)lay.V;
    if (int rc = launch_exit(*d, lay, packed, ws + w.h2, a, st)) return rc;
    if (concat_out) {   // the workspace holds split-pair rows; hand back plain fp32
        hipLaunchKernelGGL(unsplit_concat_kernel, dim3(1024), dim3(256), 0, st, ws + w.concat, concat_out, w.rows, w.rows_pad, packed, lay);
        LAUNCH_CHECK();
    }
    return 0;
}

// workspace of dnag_pred_forward_grad: time embeddings [rows][144] (rows = batch, or 1 for a shared time), then c0 [hs]
extern "C" size_t dnag_pred_workspace_bytes(const dnag_pred_dims *d, int batch) {
    if (!d || batch < 1) return 0;
    return (up((size_t)batch * d->gcn_hidden, 64) + up((size_t)d->hs, 64)) * sizeof(float);
}

extern "C" int dnag_pred_forward_grad(const dnag_pred_dims *d, const float *packed, const float *x, const float *time_cond,
                                      int time_stride, const float *mask, int mask_stride, const float *d_mu, float *y,
                                      float *grad, int batch, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_pred(d)) return rc;
    if (!packed || !x || !time_cond || !mask || !d_mu || !workspace) return fail(DNAG_E_ARG, "NULL pointer");
    if (batch < 1) return fail(DNAG_E_ARG, "batch=%d", batch);
    const PredLayout lay = pred_layout(*d);
    const size_t nt = time_stride ? batch : 1;
    const size_t need = dnag_pred_workspace_bytes(d, time_stride ? batch : 1);
    if (workspace_bytes < need) return fail(DNAG_E_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    float *temb = (float *)workspace, *c0 = temb + up(nt * lay.g, 64);
    if (int rc = launch_time_embed(time_cond, (int)nt, lay.g, packed, lay.freq, lay.t1_w, lay.t1_b, lay.t2_w, lay.t2_b, temb, st)) return rc;
    if (int rc = launch_pred_c0(lay, packed, d_mu, c0, st)) return rc;
    return launch_predictor(lay, packed, x, temb, time_stride ? 1 : 0, mask, mask_stride ? 1 : 0, c0, y, grad, batch, st);
}

// ---------------------------------------------------------------------------------------------
// dnag_pred_forward_grad without a workspace argument and without any allocation (include/dnag.h): the time-embedding
// staging that dnag_pred_forward_grad keeps in the caller's workspace lives in a file-scope device array instead.
// `scratch` is a host-side stand-in for the symbol: the device address of each GPU's copy comes from
// hipGetSymbolAddress, once per device, and only that pointer is ever handed to a kernel.
// ---------------------------------------------------------------------------------------------
#define NOALLOC_MAX_BATCH DNAG_NOALLOC_MAX_BATCH
static __device__ __align__(256) float scratch[(size_t)NOALLOC_MAX_BATCH * PRED_GH + PRED_MAXHS];   // temb [batch][144] | c0 [hs]: 2.25 MiB per GPU
static float *g_scratch_ptr[64];
static std::mutex g_scratch_mu;

static int scratch_for_current_device(float **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(DNAG_E_ARG, "device ordinal %d is beyond the 64 this library keeps scratch pointers for", dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float *p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(DNAG_E_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", dev);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int dnag_pred_forward_grad_noalloc(const dnag_pred_dims *d, const float *packed, const float *x, const float *time_cond,
                                              int time_stride, const float *mask, int mask_stride, const float *d_mu, float *y,
                                              float *grad, int batch, void *stream) {
    if (int rc = check_pred(d)) return rc;
    if (!packed || !x || !time_cond || !mask || !d_mu) return fail(DNAG_E_ARG, "NULL pointer");
    if (batch < 1) return fail(DNAG_E_ARG, "batch=%d", batch);
    if (batch > NOALLOC_MAX_BATCH)
        return fail(DNAG_E_ARG, "batch=%d exceeds DNAG_NOALLOC_MAX_BATCH=%d of the allocation-free entry point; use dnag_pred_forward_grad with a workspace",
                    batch, NOALLOC_MAX_BATCH);
    float *temb = nullptr;
    if (int rc = scratch_for_current_device(&temb)) return rc;
    const PredLayout lay = pred_layout(*d);
    const size_t nt = time_stride ? batch : 1;
    hipStream_t st = (hipStream_t)stream;
    // every temb row the predictor kernel reads (nt rows) and all of c0 are rewritten by these launches: nothing of an
    // earlier call survives
    float *c0 = temb + (size_t)NOALLOC_MAX_BATCH * PRED_GH;
    if (int rc = launch_time_embed(time_cond, (int)nt, lay.g, packed, lay.freq, lay.t1_w, lay.t1_b, lay.t2_w, lay.t2_b, temb, st)) return rc;
    if (int rc = launch_pred_c0(lay, packed, d_mu, c0, st)) return rc;
    return launch_predictor(lay, packed, x, temb, time_stride ? 1 : 0, mask, mask_stride ? 1 : 0, c0, y, grad, batch, st);
}

// ---------------------------------------------------------------------------------------------
// MetaPredictorCATE (OFA surrogate): forward and input gradient as a sequence of launches
// ---------------------------------------------------------------------------------------------
#define PC_N 20
#define PC_V 9
#define PC_G 3

// once per call: c0 = b0 + W0[:, :nz] D_mu, and the zero K-padding of the flattened activations
static int predcate_prepare(const PredCateLayout &y, const float *P, const float *d_mu, float *ws, const PcWs &w, hipStream_t st) {
    hipLaunchKernelGGL(predfc0_bias_kernel, dim3((y.hs + 255) / 256), dim3(256), 0, st, P + y.p0_wd, P + y.p0_b, d_mu, y.hs, y.nz,
                       ws + w.beff);
    LAUNCH_CHECK();
    HIP_TRY(hipMemsetAsync(ws + w.flat, 0, w.bpad * y.RK * sizeof(float), st));
    // operand scales of the two final-MLP weight matrices: constants of the call (slots 0, 1 of amax; run_predcate clears 2..7)
    const CateLayout &e = y.enc;
    HIP_TRY(hipMemsetAsync(ws + w.amax, 0, 2 * sizeof(float), st));
    hipLaunchKernelGGL(amax_kernel, dim3(64), dim3(256), 0, st, P + e.f0_w, (size_t)e.Hd * e.F, ws + w.amax);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(amax_kernel, dim3(64), dim3(256), 0, st, P + e.f1_w, (size_t)e.Hd * e.Hd, ws + w.amax + 1);
    LAUNCH_CHECK();
    return 0;
}

// x [batch][20][9] -> y [batch] (or null), grad [batch][20][9] (or null); temb rows of 64
static int run_predcate(const PredCateLayout &y, const float *P, const float *x, const float *temb, int tstride, const float *mask,
                        int mstride, float *yout, float *grad, int batch, float *ws, const PcWs &w, hipStream_t st) {
    const CateLayout &e = y.enc;
    const int F = e.F, Hd = e.Hd;
    EncSave sv;
    sv.h = ws + w.sv_h; sv.h1 = ws + w.sv_h1; sv.qkv = ws + w.sv_qkv; sv.f = ws + w.sv_f; sv.rows = w.rows_pad;
    const bool split_ok = e.Hd % SplitGeom::BN == 0 && e.F % GEMM_BK == 0;
    sv.amax_concat = split_ok ? ws + w.amax + 2 : nullptr;      // the encoder leaves max |concat| there (operand scale of the first GEMM)
    if (split_ok) HIP_TRY(hipMemsetAsync(ws + w.amax + 2, 0, 6 * sizeof(float), st));      // slots 0, 1: predcate_prepare
    {   // encoder, saving what the backward pass needs
        typedef EncGeom<PC_N, PC_G, 32> Ge;
        const size_t lds = Ge::lds_floats * sizeof(float);
        auto kern = encoder_kernel<PC_N, PC_V, PC_G, 2, 32, false, true, false, false, true>;   // saved activations, split-arithmetic linears
        static LdsAttr attr;
        if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), lds)) return rc;
        DropMasks nodrop;
        memset(&nodrop, 0, sizeof(nodrop));
        hipLaunchKernelGGL(kern, dim3((batch + PC_G - 1) / PC_G), dim3(256), lds, st, P, e, x, temb, tstride, mask, mstride,
                           ws + w.concat, (float *)nullptr, batch, sv, nodrop);
        LAUNCH_CHECK();
    }
    // final MLP (cate.py:323), flatten, regeress MLP (:325-326), pred_fc (:348)
    // The two 81 920-row GEMMs of the final MLP (and its ELU-gradient GEMM below) on the split-arithmetic fp16 MFMA kernel,
    // fp32 operands split on the fly: every operand's power-of-two scale comes from its largest magnitude, measured on
    // the device (amax_kernel, or the producing GEMM's epilogue), as in the training step.  Nout = Hd = 256 is one tile
    // column of that kernel; the GEMMs with 128 or 384 output columns stay on the fp32 MFMA kernel.
    float *amax_w0 = ws + w.amax, *amax_w1 = ws + w.amax + 1, *amax_cc = ws + w.amax + 2, *amax_h1 = ws + w.amax + 3, *amax_dh2 = ws + w.amax + 4;
    if (split_ok) {
        if (w.rows_pad > w.rows) HIP_TRY(hipMemsetAsync(ws + w.concat + w.rows * F, 0, (w.rows_pad - w.rows) * F * sizeof(float), st));
        if (int rc = launch_gemm32<ACT_ELU, 5>(ws + w.concat, F, amax_cc, P + e.f0_w, F, amax_w0, P + e.f0_b, ws + w.h1, Hd, nullptr, w.rows_pad, Hd, F, 1, st, amax_h1)) return rc;
        if (int rc = launch_gemm32<ACT_ELU, 5>(ws + w.h1, Hd, amax_h1, P + e.f1_w, Hd, amax_w1, P + e.f1_b, ws + w.h2, Hd, nullptr, w.rows_pad, Hd, Hd, 1, st)) return rc;
    } else {
        if (int rc = launch_gemm_t<ACT_ELU, 2>(ws + w.concat, P + e.f0_w, P + e.f0_b, ws + w.h1, nullptr, w.rows_pad, Hd, F, F, Hd, Hd, st)) return rc;
        if (int rc = launch_gemm_t<ACT_ELU, 2>(ws + w.h1, P + e.f1_w, P + e.f1_b, ws + w.h2, nullptr, w.rows_pad, Hd, Hd, Hd, Hd, Hd, st)) return rc;
    }
    {
        ExitArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = EXIT_FORWARD; a.batch = batch; a.out = ws + w.flat; a.out_gstride = y.RK;
        hipLaunchKernelGGL((head_out_kernel<PC_N, PC_V, PC_G>), dim3((batch + PC_G - 1) / PC_G), dim3(256), 0, st, ws + w.h2, P + e.f2_w,
                           P + e.f2_b, Hd, a);
        LAUNCH_CHECK();
    }
    if (int rc = launch_gemm_t<ACT_ELU, 2>(ws + w.flat, P + y.r0_w, P + y.r0_b, ws + w.r1, nullptr, w.bpad, y.R2P, y.RK, y.RK, y.R2P, y.R2P, st)) return rc;
    if (int rc = launch_gemm_t<ACT_NONE, 2>(ws + w.r1, P + y.r1_w, P + y.r1_b, ws + w.r2, nullptr, w.bpad, y.R2P, y.R2P, y.R2P, y.R2P, y.R2P, st)) return rc;
    if (int rc = launch_gemm_t<ACT_TANH, 2>(ws + w.r2, P + y.p0_wa, ws + w.beff, ws + w.pz, nullptr, w.bpad, y.hs, y.R2P, y.R2P, y.hs, y.hs, st)) return rc;
    hipLaunchKernelGGL(predfc2_kernel, dim3(batch), dim3(256), 0, st, ws + w.pz, y.hs, P + y.p2_w, P + y.p2_b, yout, grad ? 1 : 0);
    LAUNCH_CHECK();
    if (!grad) return 0;
    // backward: transposed weights, elu' from the stored activations
    if (int rc = launch_gemm_t<ACT_NONE, 2>(ws + w.pz, P + y.p0T, nullptr, ws + w.dr2, nullptr, w.bpad, y.R2P, y.hs, y.hs, y.R2P, y.R2P, st)) return rc;
    if (int rc = launch_gemm_t<ACT_MUL_ELUGRAD, 2>(ws + w.dr2, P + y.r1T, nullptr, ws + w.dr1, ws + w.r1, w.bpad, y.R2P, y.R2P, y.R2P, y.R2P, y.R2P, st)) return rc;
    if (int rc = launch_gemm_t<ACT_NONE, 2>(ws + w.dr1, P + y.r0T, nullptr, ws + w.dflat, nullptr, w.bpad, y.RNP, y.R2P, y.R2P, y.RK, y.RK, st)) return rc;
    {
        const size_t rows = w.rows;
        const int grid = (int)(rows < 4096 ? rows : 4096);
        hipLaunchKernelGGL((final2_bwd_kernel<PC_N, PC_V>), dim3(grid), dim3(256), 0, st, ws + w.h2, ws + w.dflat, y.RK, P + e.f2_w, Hd, rows);
        LAUNCH_CHECK();
    }
    if (split_ok) {      // dH1 = (dH2 W1) * elu'(H1); f1T holds W1's values: the same largest magnitude
        hipLaunchKernelGGL(amax_kernel, dim3(1024), dim3(256), 0, st, ws + w.h2, w.rows * (size_t)Hd, amax_dh2);
        LAUNCH_CHECK();
        if (int rc = launch_gemm32<ACT_NONE, 5>(ws + w.h2, Hd, amax_dh2, P + y.f1T, Hd, amax_w1, nullptr, ws + w.dh1, Hd, nullptr, w.rows_pad, Hd, Hd, 1, st)) return rc;
        hipLaunchKernelGGL(mul_elugrad_kernel, dim3(2048), dim3(256), 0, st, ws + w.dh1, ws + w.h1, w.rows_pad * (size_t)Hd / 4, (float *)nullptr);
        LAUNCH_CHECK();
    } else if (int rc = launch_gemm_t<ACT_MUL_ELUGRAD, 2>(ws + w.h2, P + y.f1T, nullptr, ws + w.dh1, ws + w.h1, w.rows_pad, Hd, Hd, Hd, Hd, Hd, st)) return rc;
    if (int rc = launch_gemm_t<ACT_NONE, 2>(ws + w.dh1, P + y.f0T, nullptr, ws + w.dconcat, nullptr, w.rows_pad, F, Hd, Hd, F, F, st)) return rc;
    {
        typedef PcbGeom<PC_N, PC_G, 2, 32> Ge;
        const size_t lds = Ge::lds_floats * sizeof(float);
        auto kern = predcate_enc_bwd_kernel<PC_N, PC_V, PC_G, 2, 32>;
        static LdsAttr attr;
        if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3((batch + PC_G - 1) / PC_G), dim3(256), lds, st, P, y, sv, ws + w.dconcat, mask, mstride, grad, batch);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dnag_predcate_forward_grad(const dnag_predcate_dims *d, const float *packed, const float *x, const float *time_cond,
                                          int time_stride, const float *mask, int mask_stride, const float *d_mu, float *y,
                                          float *grad, int batch, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_predcate(d)) return rc;
    if (!packed || !x || !time_cond || !mask || !d_mu || !workspace) return fail(DNAG_E_ARG, "NULL pointer");
    if (batch < 1) return fail(DNAG_E_ARG, "batch=%d", batch);
    const PredCateLayout lay = predcate_layout(*d);
    const PcWs w = predcate_ws(lay, batch);
    const size_t nt = time_stride ? batch : 1;
    const size_t temb_floats = up((size_t)batch * lay.enc.d, 64);
    if (workspace_bytes < (temb_floats + w.total) * sizeof(float))
        return fail(DNAG_E_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, (temb_floats + w.total) * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float *temb = (float *)workspace, *ws = temb + temb_floats;
    const CateLayout &e = lay.enc;
    if (int rc = launch_time_embed(time_cond, (int)nt, e.d, packed, e.freq, e.t1_w, e.t1_b, e.t2_w, e.t2_b, temb, st)) return rc;
    if (int rc = predcate_prepare(lay, packed, d_mu, ws, w, st)) return rc;
    return run_predcate(lay, packed, x, temb, time_stride ? 1 : 0, mask, mask_stride ? 1 : 0, y, grad, batch, ws, w, st);
}

// ---------------------------------------------------------------------------------------------
// predictor-corrector loop
// ---------------------------------------------------------------------------------------------
#define LABEL_CHUNK 256
struct LabelChunk { float v[LABEL_CHUNK]; };
__global__ void put_labels_kernel(float *__restrict__ dst, const LabelChunk c, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = c.v[threadIdx.x];
}

// The surrogate's gradient and the score network's forward of one evaluation read the same x and meet only in the exit
// kernel, so the surrogate runs on a second stream of the library's own (one per device, created on first use, kept until
// the process ends), forked from and joined to the caller's stream by events: its workgroups fill the CUs the other
// kernels' last rounds leave idle.  Results do not change (same kernels, disjoint workspace regions).  The first guided
// call on a device creates the stream and its two events: make it outside a stream capture.
// DNAG_OVERLAP=0 in the environment keeps everything on the caller's stream.
struct SideStream { hipStream_t st; hipEvent_t fork, join, mid; bool ok; std::mutex use; };
static SideStream g_side[64];
static std::mutex g_side_mu;
static int side_for_current_device(SideStream **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(DNAG_E_ARG, "device %d", dev);
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream &s = g_side[dev];
    if (!s.ok) {
        // highest priority: the surrogate's many short, under-filled kernels are dispatched as they become ready and the
        // score network's large ones take what is left (OFA, B = 4096, same box: 657 samples/s on one stream, 683 with
        // this stream at the lowest priority, 720 at the default, 734 at the highest)
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&s.st, hipStreamNonBlocking, hi));
        // no system-scope fence at the events: both streams are on this device (a record costs 4.8 instead of 6.4 us on the
        // recording stream; NB201 2794 -> 2809 samples/s same box)
        HIP_TRY(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming | hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&s.join, hipEventDisableTiming | hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&s.mid, hipEventDisableTiming | hipEventDisableSystemFence));
        s.ok = true;
    }
    *out = &s;
    return 0;
}
static bool overlap_enabled() {
    const char *e = getenv("DNAG_OVERLAP");      // read per call: a test compares both settings in one process
    return !(e && e[0] == '0');
}

extern "C" int dnag_pc_sample(const dnag_pc_args *p) {
    if (!p) return fail(DNAG_E_ARG, "args is NULL");
    if (int rc = check_cate(p->cate_dims)) return rc;
    const bool guided_pc = p->predcate_dims != nullptr;
    const bool guided = p->pred_dims != nullptr || guided_pc;
    if (p->pred_dims && guided_pc) return fail(DNAG_E_ARG, "pred_dims and predcate_dims are both set: pass one surrogate");
    if (guided_pc) {
        if (int rc = check_predcate(p->predcate_dims)) return rc;
        if (!p->predcate_packed || !p->d_mu) return fail(DNAG_E_ARG, "guided sampling needs predcate_packed and d_mu");
        if (p->predcate_dims->max_node != p->cate_dims->max_node || p->predcate_dims->n_vocab != p->cate_dims->n_vocab)
            return fail(DNAG_E_ARG, "score net and surrogate disagree on the graph shape");
        if (!(p->label_sign == 1.0f || p->label_sign == -1.0f)) return fail(DNAG_E_ARG, "label_sign must be +1 or -1");
    } else if (guided) {
        if (int rc = check_pred(p->pred_dims)) return rc;
        if (!p->pred_packed || !p->d_mu) return fail(DNAG_E_ARG, "guided sampling needs pred_packed and d_mu");
        if (p->pred_dims->max_node != p->cate_dims->max_node || p->pred_dims->n_vocab != p->cate_dims->n_vocab)
            return fail(DNAG_E_ARG, "score net and surrogate disagree on the graph shape");
        if (!(p->label_sign == 1.0f || p->label_sign == -1.0f)) return fail(DNAG_E_ARG, "label_sign must be +1 or -1");
    }
    if (!p->cate_packed || !p->mask || !p->sched || !p->x_out || !p->workspace) return fail(DNAG_E_ARG, "NULL pointer");
    if (p->batch < 1 || p->n_steps < 1) return fail(DNAG_E_ARG, "batch=%d n_steps=%d", p->batch, p->n_steps);
    if (p->predictor < DNAG_PREDICTOR_NONE || p->predictor > DNAG_PREDICTOR_EULER_MARUYAMA)
        return fail(DNAG_E_ARG, "predictor=%d is not a DNAG_PREDICTOR_* value", p->predictor);
    if (p->score_mode != DNAG_SCORE_VP && p->score_mode != DNAG_SCORE_VE) return fail(DNAG_E_ARG, "score_mode=%d", p->score_mode);
    if (!p->x0 && !(p->prior_scale > 0.f)) return fail(DNAG_E_ARG, "prior_scale=%g must be positive when x0 is NULL", p->prior_scale);
    if (p->weight_ratio < 0 || p->weight_ratio > 2 || (p->weight_ratio == 1 && !(p->weight_ratio_p > 0.f)))
        return fail(DNAG_E_ARG, "weight_ratio=%d weight_ratio_p=%g", p->weight_ratio, p->weight_ratio_p);
    if ((p->noise_corr == nullptr) != (p->noise_pred == nullptr) && p->corrector)
        return fail(DNAG_E_ARG, "noise_corr and noise_pred must both be given (tape mode) or both NULL (Philox)");
    const int chain_n = p->chain_n > 0 ? p->chain_n : p->n_steps;
    if (p->chain_x && (p->chain_steps < 1 || p->chain_steps > chain_n || p->chain_keep < 1 || p->chain_keep > p->batch || p->n_steps > chain_n))
        return fail(DNAG_E_ARG, "chain_steps=%d chain_keep=%d chain_n=%d (need 1 <= steps <= chain_n, 1 <= keep <= batch, n_steps <= chain_n)",
                    p->chain_steps, p->chain_keep, chain_n);
    const dnag_cate_dims &cd = *p->cate_dims;
    const int CS = p->corrector_steps > 0 ? p->corrector_steps : 1;
    if (CS > 1024) return fail(DNAG_E_ARG, "corrector_steps=%d", p->corrector_steps);
    if (p->langevin_exchange) {
        if (!p->langevin_local || !p->langevin_global || p->langevin_world < 1)
            return fail(DNAG_E_ARG, "langevin_exchange needs langevin_local, langevin_global and langevin_world >= 1");
        if (p->batch % exit_gf(cd))
            return fail(DNAG_E_ARG, "exact-global-batch Langevin mode: the shard holds %d graphs, not a multiple of %d", p->batch, exit_gf(cd));
    }
    const CateLayout lay = cate_layout(cd);
    const WsLayout w = ws_layout(cd, p->pred_dims, p->batch, p->n_steps, p->predcate_dims);
    if (p->workspace_bytes < w.total * sizeof(float))
        return fail(DNAG_E_WORKSPACE, "workspace %zu bytes < %zu needed", p->workspace_bytes, w.total * sizeof(float));
    hipStream_t st = (hipStream_t)p->stream;
    float *ws = (float *)p->workspace;
    const int B = p->batch, S = p->n_steps;
    const size_t nelem = (size_t)B * cd.max_node * cd.n_vocab;
    PredLayout play;
    PredCateLayout pclay;
    PcWs pcw;
    if (guided_pc) { pclay = predcate_layout(*p->predcate_dims); pcw = predcate_ws(pclay, B); }
    else if (guided) play = pred_layout(*p->pred_dims);

    // labels -> device, time embeddings of every step in one launch per network
    std::vector<float> labels(S);
    for (int i = 0; i < S; ++i) {
        const float *row = p->sched + (size_t)i * DNAG_SCHED_COLS;
        if (p->score_mode == DNAG_SCORE_VP && !(row[DNAG_SCHED_STD] > 0.f))
            return fail(DNAG_E_ARG, "sched[%d].std = %g is not positive", i, row[DNAG_SCHED_STD]);
        labels[i] = row[DNAG_SCHED_LABEL];
    }
    // the labels travel as kernel arguments (copied at launch): no host buffer has to outlive this call, nothing waits
    for (int i = 0; i < S; i += LABEL_CHUNK) {
        LabelChunk ch;
        const int n = S - i < LABEL_CHUNK ? S - i : LABEL_CHUNK;
        memcpy(ch.v, labels.data() + i, n * sizeof(float));
        hipLaunchKernelGGL(put_labels_kernel, dim3(1), dim3(LABEL_CHUNK), 0, st, ws + w.labels + i, ch, n);
        LAUNCH_CHECK();
    }
    if (int rc = launch_time_embed(ws + w.labels, S, lay.d, p->cate_packed, lay.freq, lay.t1_w, lay.t1_b, lay.t2_w, lay.t2_b, ws + w.temb_s, st)) return rc;
    if (guided_pc) {
        const CateLayout &e = pclay.enc;
        if (int rc = launch_time_embed(ws + w.labels, S, e.d, p->predcate_packed, e.freq, e.t1_w, e.t1_b, e.t2_w, e.t2_b, ws + w.temb_p, st)) return rc;
        if (int rc = predcate_prepare(pclay, p->predcate_packed, p->d_mu, ws + w.pc, pcw, st)) return rc;
    } else if (guided) {
        if (int rc = launch_time_embed(ws + w.labels, S, play.g, p->pred_packed, play.freq, play.t1_w, play.t1_b, play.t2_w, play.t2_b, ws + w.temb_p, st)) return rc;
        if (int rc = launch_pred_c0(play, p->pred_packed, p->d_mu, ws + w.c0, st)) return rc;
    }

    float *x = ws + w.x, *x_mean = ws + w.x_mean, *cgrad = ws + w.cgrad, *gbuf = ws + w.gbuf, *partials = ws + w.partials;
    hipLaunchKernelGGL(prior_kernel, dim3(1024), dim3(256), 0, st, x, p->x0, nelem, p->seed, p->stream_offset, p->prior_scale);
    LAUNCH_CHECK();

    auto rec_at = [&](float *base, int i) -> float * { return base ? base + (size_t)i * nelem : nullptr; };
    SideStream *side = nullptr;
    if (guided && overlap_enabled()) if (int rc = side_for_current_device(&side)) return rc;
    // one trajectory at a time enqueues on a device's second stream: a wait takes the event's record as of the call, so
    // two host threads interleaving their records of the same pair of events would wait for each other's work
    std::unique_lock<std::mutex> side_lock;
    if (side) side_lock = std::unique_lock<std::mutex>(side->use);

    for (int i = 0; i < S; ++i) {
        const float *row = p->sched + (size_t)i * DNAG_SCHED_COLS;
        // sub-steps 0 .. CS-1: corrector updates (sampling.py:244, :283-293), sub-step CS: predictor; a None corrector /
        // predictor evaluates nothing (sampling.py:216-224, :259-267)
        for (int sub = p->corrector ? 0 : CS; sub < (p->predictor == DNAG_PREDICTOR_NONE ? CS : CS + 1); ++sub) {
            const bool corr = sub < CS;
            const int ci = i * CS + sub;          // index of this corrector update in the tape / the records
            const bool guide_now = guided && (p->guide_steps == nullptr || p->guide_steps[i] != 0);
            if (guide_now) {
                hipStream_t sg = st;
                if (side) { HIP_TRY(hipEventRecord(side->fork, st)); HIP_TRY(hipStreamWaitEvent(side->st, side->fork, 0)); sg = side->st; }
                if (guided_pc) {
                    if (int rc = run_predcate(pclay, p->predcate_packed, x, ws + w.temb_p + (size_t)i * pclay.enc.d, 0, p->mask, 0, nullptr, cgrad, B, ws + w.pc, pcw, sg)) return rc;
                } else if (int rc = launch_predictor(play, p->pred_packed, x, ws + w.temb_p + (size_t)i * play.g, 0, p->mask, 0, ws + w.c0, nullptr, cgrad, B, sg)) return rc;
                float *rc_ = corr ? rec_at(p->rec_cgrad_c, ci) : rec_at(p->rec_cgrad_p, i);
                if (rc_) HIP_TRY(hipMemcpyAsync(rc_, cgrad, nelem * sizeof(float), hipMemcpyDeviceToDevice, sg));
                if (side) HIP_TRY(hipEventRecord(side->join, side->st));
            }
            if (int rc = run_score_body(cd, lay, p->cate_packed, x, ws + w.temb_s + (size_t)i * lay.d, 0, p->mask, 0, ws, w, nullptr, B, st)) return rc;
            if (guide_now && side) HIP_TRY(hipStreamWaitEvent(st, side->join, 0));
            ExitArgs a;
            memset(&a, 0, sizeof(a));
            a.mode = corr ? EXIT_CORRECTOR : EXIT_PREDICTOR;
            a.batch = B;
            a.std = row[DNAG_SCHED_STD];
            a.score_mode = p->score_mode; a.pred_kind = p->predictor; a.dt = row[DNAG_SCHED_DT];
            a.cgrad = guide_now ? cgrad : nullptr;
            a.cscale = (p->cscale_steps ? p->cscale_steps[i] : p->classifier_scale) * p->label_sign;
            a.wr_mode = p->weight_ratio; a.wr_p = p->weight_ratio_p; a.prod_w = p->prod_w;
            a.gbuf = gbuf; a.partials = p->langevin_exchange ? p->langevin_local : partials;
            a.x = x; a.x_mean = x_mean;
            a.sqrt_alpha = row[DNAG_SCHED_SQRTA]; a.G = row[DNAG_SCHED_G]; a.G2 = row[DNAG_SCHED_G2];
            const float *tape = corr ? p->noise_corr : p->noise_pred;
            a.noise = tape ? tape + (size_t)(corr ? ci : i) * nelem : nullptr;
            a.seed = p->seed; a.stream_offset = p->stream_offset; a.draw = (uint32_t)(1 + i * (CS + 1) + sub);
            a.rec_score = corr ? rec_at(p->rec_score_c, ci) : rec_at(p->rec_score_p, i);
            a.rec_total = corr ? rec_at(p->rec_total_c, ci) : rec_at(p->rec_total_p, i);
            a.rec_a = corr ? nullptr : rec_at(p->rec_x_mean, i);
            a.rec_b = corr ? nullptr : rec_at(p->rec_x_pred, i);
            a.parts = ws + w.h2; a.nparts = lay.Hd / SplitGeom::BN; a.part_stride = w.rows_pad * (size_t)lay.V;
            if (int rc = launch_exit(cd, lay, p->cate_packed, ws + w.h2, a, st)) return rc;
            if (corr) {
                CorrArgs c;
                memset(&c, 0, sizeof(c));
                c.partials = partials; c.nparts = w.nblk_exit; c.nseg = 1; c.batch = B; c.nelem = nelem;
                if (p->langevin_exchange) {
                    // every rank's per-block sums, in rank order, reduced as one batch of world * B graphs would be
                    if (int rc = p->langevin_exchange(p->langevin_user, p->stream))
                        return fail(DNAG_E_ARG, "langevin_exchange callback failed with %d at PC step %d", rc, i);
                    c.partials = p->langevin_global; c.nseg = p->langevin_world; c.nparts = w.nblk_exit * p->langevin_world;
                    c.batch = B * p->langevin_world;
                }
                c.snr = p->snr; c.alpha = row[DNAG_SCHED_ALPHA];
                c.gbuf = gbuf; c.x = x;
                c.noise = a.noise; c.seed = a.seed; c.stream_offset = a.stream_offset; c.draw = a.draw;
                c.rec_x = rec_at(p->rec_x_corr, ci);
                c.rec_step = p->rec_step_size ? p->rec_step_size + (size_t)ci * 3 : nullptr;
                const int grid = (int)((nelem + 255) / 256 < 2048 ? (nelem + 255) / 256 : 2048);
                hipLaunchKernelGGL(corrector_update_kernel, dim3(grid), dim3(256), 0, st, c);
                LAUNCH_CHECK();
            }
        }
        if (p->chain_x) {   // MobileNetV3/sampling.py:1030-1034; only the last step of a slot survives there
            const long long cs = p->chain_steps;
            const long long slot = (i * cs) / chain_n;
            if (i == S - 1 || ((i + 1) * cs) / chain_n != slot) {
                // a None predictor returns (x, x): x_mean is x then (sampling.py:216-224)
                const float *src = (p->chain_denoise && p->predictor != DNAG_PREDICTOR_NONE) ? x_mean : x;
                const size_t n = (size_t)p->chain_keep * cd.max_node * cd.n_vocab;
                HIP_TRY(hipMemcpyAsync(p->chain_x + (size_t)(cs - 1 - slot) * n, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
        }
    }
    // NonePredictor returns (x, x): the state after the last corrector (or the prior) is also x_mean
    if (p->predictor == DNAG_PREDICTOR_NONE) HIP_TRY(hipMemcpyAsync(x_mean, x, nelem * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(inverse_scale_kernel, dim3(1024), dim3(256), 0, st, x_mean, p->x_out, nelem);
    LAUNCH_CHECK();
    if (p->x_last) HIP_TRY(hipMemcpyAsync(p->x_last, x, nelem * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (p->x_mean_last) HIP_TRY(hipMemcpyAsync(p->x_mean_last, x_mean, nelem * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}


// ---------------------------------------------------------------------------------------------
// DSM training step
// ---------------------------------------------------------------------------------------------
struct TrainWs {
    size_t rows, rows_pad;
    int nblk_enc, nz_rows, nz_b;
    size_t packed, temb, save_e, save_pre, dpre, dtau, xt, concat, h1, h2, out, dout, dh2, dh1, dconcat, T1, T2;
    size_t sv_h, sv_h1, sv_qkv, sv_f, part, gsum, scratch, splitk, lossp, amax, total;
};
#define TRAIN_LOSS_BLOCKS 128
// graphs per workgroup of the encoder BACKWARD: 8 (64 rows, best MFMA tile reuse) once that fills the chip, else 4
// (32 rows: twice the workgroups at 2/3 of the latency; measured 1077 -> 712 us at B = 1024).  The forward keeps 8:
// its latency is set by the weight-fragment loads, which do not shrink with the rows (496 us at G = 8, 589 at G = 4).
static int train_g(int batch) { return batch >= 2048 ? 8 : 4; }

static TrainWs train_ws(const dnag_cate_dims &c, int batch) {
    TrainWs w;
    const TrainLayout tl = train_layout(c);
    const GradLayout gl = grad_layout(c);
    const size_t N = c.max_node, V = c.n_vocab, d = c.d_model, dff = c.d_ff, L = c.n_layers, F = tl.enc.F, Hd = tl.enc.Hd;
    w.rows = (size_t)batch * N;
    w.rows_pad = up(w.rows, SplitGeom::BM);
    w.nblk_enc = (batch + train_g(batch) - 1) / train_g(batch);
    w.nz_rows = (int)((w.rows + 127) / 128 < 64 ? (w.rows + 127) / 128 : 64);
    w.nz_b = (batch + 15) / 16 < 64 ? (batch + 15) / 16 : 64;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += up(n, 64); return at; };
    w.packed = take(tl.total);
    w.temb = take((size_t)batch * d);
    w.save_e = take((size_t)batch * d);
    w.save_pre = take((size_t)batch * 4 * d);
    w.dpre = take((size_t)batch * 4 * d);
    w.dtau = take((size_t)batch * d);
    w.xt = take(w.rows * V);
    w.concat = take(w.rows_pad * F);
    w.h1 = take(w.rows_pad * Hd);
    w.h2 = take(w.rows_pad * Hd);
    w.out = take(w.rows_pad * V);
    w.dout = take(w.rows_pad * V);
    w.dh2 = take(w.rows_pad * Hd);
    w.dh1 = take(w.rows_pad * Hd);
    w.dconcat = take(w.rows_pad * F);
    w.T1 = take(Hd * w.rows_pad);
    w.T2 = take(Hd * w.rows_pad);
    w.sv_h = take((L + 1) * w.rows_pad * d);
    w.sv_h1 = take(L * w.rows_pad * d);
    w.sv_qkv = take(L * w.rows_pad * 3 * d);
    w.sv_f = take(L * w.rows_pad * dff);
    w.part = take((size_t)w.nblk_enc * gl.total);
    w.gsum = take(gl.total);
    const size_t big = (V * Hd > 4 * d * d ? V * Hd : 4 * d * d);
    w.scratch = take(64 * big);
    w.splitk = take(8 * Hd * Hd);
    w.lossp = take(TRAIN_LOSS_BLOCKS);
    w.amax = take(8);            // max |dH2|, |dH1|, |W0|, |W1|, |concat|, |H1| (float bits): dynamic operand scales of the split GEMMs
    w.total = o;
    return w;
}

static int check_train_dims(const dnag_cate_dims *d) {
    if (int rc = check_cate(d)) return rc;
    if (!is_nb201(*d)) return fail(DNAG_E_ARG, "the training step is built for NAS-Bench-201 graphs (8x7) only");
    return 0;
}

extern "C" size_t dnag_train_workspace_bytes(const dnag_cate_dims *d, int batch) {
    if (!d || batch < 1) return 0;
    return train_ws(*d, batch).total * sizeof(float);
}
extern "C" size_t dnag_dropout_mask_bytes(const dnag_cate_dims *d, int batch) {
    if (!d || batch < 1) return 0;
    const size_t rows = (size_t)batch * d->max_node;
    const size_t per_layer = (size_t)d->n_head * d->max_node + 2 * (size_t)d->d_model + d->d_ff;
    return rows * d->d_model + (size_t)d->n_layers * rows * per_layer;
}
extern "C" int dnag_dropout_fill(const dnag_cate_dims *d, int batch, float p, uint64_t seed, uint8_t *keep, void *stream) {
    if (int rc = check_cate(d)) return rc;
    if (!keep || batch < 1 || !(p >= 0.f && p < 1.f)) return fail(DNAG_E_ARG, "bad dropout arguments");
    const size_t n = dnag_dropout_mask_bytes(d, batch);
    hipLaunchKernelGGL(dropout_fill_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, keep, n, p, seed);
    LAUNCH_CHECK();
    return 0;
}

// out[I][J] (row stride J, scaled) = sum over rows of fa(A[r][i]) fb(B[r][j]) via row chunks + ordered reduction
template <int ACTB>
static int launch_tn_small(const float *A, int lda, const float *Bm, int ldb, size_t rows, int nz, int I, int J, float *scratch,
                           float *out, hipStream_t st) {
    const int chunk = (int)((rows + nz - 1) / nz);
    if (ACTB == 0 && I <= 8)
        hipLaunchKernelGGL((tn_few_kernel<8>), dim3((J + 255) / 256, 1, nz), dim3(256), 0, st, A, lda, Bm, ldb, (int)rows, chunk, I, J, scratch);
    else
        hipLaunchKernelGGL((tn_small_kernel<ACTB>), dim3((I * J + 255) / 256, 1, nz), dim3(256), 0, st, A, lda, Bm, ldb, (int)rows, chunk, I, J, scratch);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((I * J + 255) / 256), dim3(256), 0, st, scratch, nz, (size_t)I * J, (size_t)I * J, out, 1.0f);
    LAUNCH_CHECK();
    return 0;
}
static int launch_colsum(const float *A, int lda, size_t rows, int nz, int J, float *scratch, float *out, hipStream_t st) {
    const int chunk = (int)((rows + nz - 1) / nz);
    hipLaunchKernelGGL(colsum_kernel, dim3((J + 255) / 256, 1, nz), dim3(256), 0, st, A, lda, (int)rows, chunk, J, scratch);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((J + 255) / 256), dim3(256), 0, st, scratch, nz, (size_t)J, (size_t)J, out, 1.0f);
    LAUNCH_CHECK();
    return 0;
}
static int launch_transpose(const float *in, size_t rows, int cols, int ld_in, float *out, size_t rows_out, size_t ld_out, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (unsigned)((rows_out + 31) / 32)), dim3(256), 0, st, in, (int)rows, cols, ld_in, out,
                       (int)rows_out, (int)ld_out);
    LAUNCH_CHECK();
    return 0;
}

template <int G>
static int launch_train_encoder(bool drop, int nblk, const float *P, const CateLayout &e, const float *xt, const float *temb, const float *mask,
                                int mstride, float *concat, int B, const EncSave &sv, const DropMasks &dm, hipStream_t st) {
    // dnag_encoder8.hip.h in its training modes (saved activations, fp32 concat with the output norm's affine, dropout);
    // the generic encoder_kernel<8, 7, G, 8, 128, true, true, DROP, false, true> it replaces took 415 us at B = 1024
    typedef Enc8Geom<8> Ge;
    static_assert(G == Ge::G, "eight graphs per workgroup");
    const size_t lds = Ge::lds_floats * sizeof(float);
    auto k_drop = encoder8_kernel<8, 7, 2>;
    auto k_nodrop = encoder8_kernel<8, 7, 1>;
    static LdsAttr a_drop, a_nodrop;
    if (int rc = ensure_lds(a_drop, reinterpret_cast<const void *>(k_drop), lds)) return rc;
    if (int rc = ensure_lds(a_nodrop, reinterpret_cast<const void *>(k_nodrop), lds)) return rc;
    hipLaunchKernelGGL(drop ? k_drop : k_nodrop, dim3(nblk), dim3(256), lds, st, P, e, xt, temb, 1, mask, mstride, concat, B, sv, dm);
    LAUNCH_CHECK();
    return 0;
}

template <int G>
static int launch_train_enc_bwd(int nblk, const float *P, const TrainLayout &tl, const EncSave &sv, const float *xt, const float *dconcat,
                                const float *mask, int mstride, const DropMasks &dm, float *part, const GradLayout &gl, float *dtau, int B,
                                hipStream_t st) {
    typedef TrbGeom<8, G, 8, 128> Ge;
    const size_t lds = Ge::lds_floats * sizeof(float);
    auto kern = cate_enc_bwd_kernel<8, 7, G, 8, 128>;
    static LdsAttr attr;
    if (int rc = ensure_lds(attr, reinterpret_cast<const void *>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, P, tl, sv, xt, dconcat, mask, mstride, dm, part, gl, dtau, B);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int dnag_cate_dsm_step(const dnag_train_args *p) {
    if (!p) return fail(DNAG_E_ARG, "args is NULL");
    if (int rc = check_train_dims(p->cate_dims)) return rc;
    if (!p->params || !p->tables || !p->x || !p->z || !p->mean_coef || !p->std || !p->labels || !p->mask || !p->loss || !p->workspace)
        return fail(DNAG_E_ARG, "NULL pointer");
    if (p->batch < 1) return fail(DNAG_E_ARG, "batch=%d", p->batch);
    if (p->score_mode != DNAG_SCORE_VP && p->score_mode != DNAG_SCORE_VE) return fail(DNAG_E_ARG, "score_mode=%d", p->score_mode);
    if (!(p->dropout_p >= 0.f && p->dropout_p < 1.f)) return fail(DNAG_E_ARG, "dropout_p=%g", p->dropout_p);
    const dnag_cate_dims &cd = *p->cate_dims;
    const CateStateLayout sl = cate_state_layout(cd);
    const TrainLayout tl = train_layout(cd);
    const CateLayout &e = tl.enc;
    const GradLayout gl = grad_layout(cd);
    const TrainWs w = train_ws(cd, p->batch);
    if (p->workspace_bytes < w.total * sizeof(float))
        return fail(DNAG_E_WORKSPACE, "workspace %zu bytes < %zu needed", p->workspace_bytes, w.total * sizeof(float));
    hipStream_t st = (hipStream_t)p->stream;
    float *ws = (float *)p->workspace;
    const float *S = p->params;
    float *P = ws + w.packed;
    const int B = p->batch, N = cd.max_node, V = cd.n_vocab, d = e.d, dff = e.dff, L = e.L, F = e.F, Hd = e.Hd;
    const size_t rows = w.rows, rows_pad = w.rows_pad;
    const bool drop = p->keep != nullptr && p->dropout_p > 0.f;
    CopySegs segs;
    segs.count = 0;
    auto cpy = [&](float *dst, const float *src, size_t n) { const int k = segs.count++; segs.src[k] = src; segs.dst[k] = dst; segs.n[k] = n; };

    // ---- 0. pack: state order -> kernel layouts (forward + transposed backward operands) -------------------
    cpy(P + e.emb_w, S + sl.emb_w, (size_t)d * V);
    cpy(P + e.emb_b, S + sl.emb_b, (size_t)d + 4 * d * d + 4 * d + 4 * d * d + d);      // emb_b, timeEmb1, timeEmb2
    cpy(P + e.f0_w, S + sl.f0_w, (size_t)Hd * F + Hd + (size_t)Hd * Hd + Hd);             // final.linears.0, .1
    cpy(P + e.f2_w, S + sl.f2_w, (size_t)V * Hd);
    cpy(P + e.f2_b, S + sl.f2_b, (size_t)V);
    cpy(P + e.pe, p->tables, (size_t)N * d);
    cpy(P + e.freq, p->tables + (size_t)N * d, (size_t)d / 2);
    hipLaunchKernelGGL(copy_segs_kernel, dim3(512), dim3(256), 0, st, segs);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(train_scales_kernel, dim3(L), dim3(256), 0, st, S, sl, P, tl, drop ? 1.0f / (1.0f - p->dropout_p) : 1.0f);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(train_pack_layers_kernel, dim3(32, L), dim3(256), 0, st, S, sl, P, tl);
    LAUNCH_CHECK();
    // The device's second stream (SideStream, above; sw = st when DNAG_OVERLAP=0) takes what hangs off the step's main
    // chain without feeding its next kernel: here the head weights' scales and backward transposes, which the chain needs
    // only after the encoder forward (128 workgroups at B = 1024: half the chip is free meanwhile).
    SideStream *side = nullptr;
    if (overlap_enabled()) if (int rc = side_for_current_device(&side)) return rc;
    std::unique_lock<std::mutex> side_lock;
    if (side) side_lock = std::unique_lock<std::mutex>(side->use);
    hipStream_t sw = side ? side->st : st;
    auto fork = [&]() -> int {
        if (side) { HIP_TRY(hipEventRecord(side->fork, st)); HIP_TRY(hipStreamWaitEvent(side->st, side->fork, 0)); }
        return 0;
    };
    auto join = [&]() -> int {
        if (side) { HIP_TRY(hipEventRecord(side->join, side->st)); HIP_TRY(hipStreamWaitEvent(st, side->join, 0)); }
        return 0;
    };
    // every operand's power-of-two scale comes from its largest magnitude, measured on the device (no range to respect)
    HIP_TRY(hipMemsetAsync(ws + w.amax, 0, 8 * sizeof(float), st));
    float *amax_dh2 = ws + w.amax, *amax_dh1 = ws + w.amax + 1, *amax_w0 = ws + w.amax + 2, *amax_w1 = ws + w.amax + 3;
    float *amax_cc = ws + w.amax + 4, *amax_h1 = ws + w.amax + 5;
    if (side) HIP_TRY(hipEventRecord(side->fork, st));      // the packed head weights and the cleared scales are in place
    // ---- 1. time embedding per graph, perturbation -----------------------------------------------------------
    hipLaunchKernelGGL(time_embed_train_kernel, dim3(B), dim3(256), (size_t)5 * d * sizeof(float), st, p->labels, d, P + e.freq, P + e.t1_w,
                       P + e.t1_b, P + e.t2_w, P + e.t2_b, ws + w.temb, ws + w.save_e, ws + w.save_pre);
    LAUNCH_CHECK();
    const size_t nelem = rows * V;
    hipLaunchKernelGGL(perturb_kernel, dim3(512), dim3(256), 0, st, p->x, p->z, p->mean_coef, p->std, ws + w.xt, nelem, N * V);
    LAUNCH_CHECK();
    if (rows_pad > rows) HIP_TRY(hipMemsetAsync(ws + w.concat + rows * F, 0, (rows_pad - rows) * F * sizeof(float), st));
    // ---- 2. encoder forward, saving the activations the backward pass needs -------------------------------------
    EncSave sv;
    sv.h = ws + w.sv_h; sv.h1 = ws + w.sv_h1; sv.qkv = ws + w.sv_qkv; sv.f = ws + w.sv_f; sv.rows = rows_pad;
    sv.amax_concat = amax_cc;          // max |concat| for the head GEMMs' operand scale, taken where the concat is written
    DropMasks dm;
    memset(&dm, 0, sizeof(dm));
    if (drop) { dm.keep = p->keep; dm.keep_prob = 1.0f - p->dropout_p; dm.rows = rows; dm.per_layer = cd.n_head * N + 2 * d + dff; }
    if (int rc = launch_train_encoder<8>(drop, (B + 7) / 8, P, e, ws + w.xt, ws + w.temb, p->mask, p->mask_stride ? 1 : 0, ws + w.concat, B, sv, dm, st))
        return rc;
    // (the event was recorded in front of section 1; the wait and these launches are enqueued behind the encoder's launch:
    // enqueued in front of it, the main stream's next kernel started only after they had run)
    if (side) HIP_TRY(hipStreamWaitEvent(side->st, side->fork, 0));
    hipLaunchKernelGGL(amax_kernel, dim3(256), dim3(256), 0, sw, P + e.f0_w, (size_t)Hd * F, amax_w0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(amax_kernel, dim3(256), dim3(256), 0, sw, P + e.f1_w, (size_t)Hd * Hd, amax_w1);
    LAUNCH_CHECK();
    if (p->grads) {
        if (int rc = launch_transpose(S + sl.f0_w, Hd, F, F, P + tl.f0T, Hd, Hd, sw)) return rc;      // [Hd][F] -> [F][Hd]
        if (int rc = launch_transpose(S + sl.f1_w, Hd, Hd, Hd, P + tl.f1T, Hd, Hd, sw)) return rc;
    }
    // ---- 3. head MLP and loss ------------------------------------------------------------------------------------
    // all six big GEMMs of the step run on the split-arithmetic fp16 MFMA kernel with fp32 operands split on the fly
    if (int rc = join()) return rc;            // the weights' scales (and transposes)
    if (int rc = launch_gemm32<ACT_ELU, 3>(ws + w.concat, F, amax_cc, P + e.f0_w, F, amax_w0, P + e.f0_b, ws + w.h1, Hd, nullptr, rows_pad, Hd, F, 1, st, amax_h1)) return rc;
    if (int rc = launch_gemm32<ACT_ELU, 3>(ws + w.h1, Hd, amax_h1, P + e.f1_w, Hd, amax_w1, P + e.f1_b, ws + w.h2, Hd, nullptr, rows_pad, Hd, Hd, 1, st)) return rc;
    if (V == 7 && Hd % 4 == 0) {       // out = H2 W3^T + b3, one wave per row
        hipLaunchKernelGGL((rowdot_kernel<7>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, ws + w.h2, P + e.f2_w, P + e.f2_b, Hd, rows, ws + w.out);
        LAUNCH_CHECK();
    } else {
        ExitArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = EXIT_FORWARD; a.batch = B; a.out = ws + w.out;
        if (int rc = launch_exit(cd, e, P, ws + w.h2, a, st)) return rc;
    }
    hipLaunchKernelGGL(dsm_loss_kernel, dim3(TRAIN_LOSS_BLOCKS), dim3(256), 0, st, ws + w.out, p->z, p->std, p->score_mode, p->reduce_mean ? 1 : 0, B,
                       N * V, rows_pad * V, ws + w.dout, ws + w.lossp);
    LAUNCH_CHECK();
    // the loss itself feeds nothing: with gradients its one-workgroup sum runs on the second stream (after the first fork)
    auto loss_sum = [&](hipStream_t s_) -> int {
        const float scale = p->reduce_mean ? 1.0f / ((float)(N * V) * (float)B) : 0.5f / (float)B;
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(1), dim3(256), 0, s_, ws + w.lossp, TRAIN_LOSS_BLOCKS, (size_t)1, (size_t)1, p->loss, scale);
        LAUNCH_CHECK();
        return 0;
    };
    if (!p->grads) return loss_sum(st);
    float *G = p->grads;
    float *scr = ws + w.scratch;
    // The head's weight and bias gradients hang off the chain dout -> dH2 -> dH1 -> dconcat -> encoder backward without
    // feeding it: they run on the second stream, forked where their operands are complete, and the step ends with the join.
    // The encoder backward leaves no CU to anything else (one workgroup per CU, and no GEMM tile fits beside it), so what
    // the second stream still holds when it starts (most of dW1) runs after it, beside the HBM-bound slab reduction; the
    // time-embedding gradients (a string of short kernels) go in front of that reduction on the main chain, where dW1's
    // GEMM fills the chip around them (behind the reduction: 517 -> 505 steps/s; on the second stream behind dW1: 512).
    // They share `scr` with the head's column sums: ordered by the `mid` event.
    // ---- 4. head backward ------------------------------------------------------------------------------------------
    hipLaunchKernelGGL((head_out_bwd_kernel<7>), dim3((unsigned)(rows_pad < 2048 ? rows_pad : 2048)), dim3(256), 0, st, ws + w.h2, ws + w.dout,
                       P + e.f2_w, Hd, rows, rows_pad, ws + w.dh2, amax_dh2);
    LAUNCH_CHECK();
    if (int rc = fork()) return rc;            // dout, dH2 and its scale are complete
    if (int rc = loss_sum(sw)) return rc;
    if (int rc = launch_tn_small<0>(ws + w.dout, V, ws + w.h2, Hd, rows, w.nz_rows, V, Hd, scr, G + sl.f2_w, sw)) return rc;     // dW3 = dout^T H2
    if (int rc = launch_colsum(ws + w.dout, V, rows, w.nz_rows, V, scr, G + sl.f2_b, sw)) return rc;
    if (int rc = launch_colsum(ws + w.dh2, Hd, rows, w.nz_rows, Hd, scr, G + sl.f1_b, sw)) return rc;
    if (int rc = launch_transpose(ws + w.dh2, rows_pad, Hd, Hd, ws + w.T1, rows_pad, rows_pad, sw)) return rc;
    if (int rc = launch_transpose(ws + w.h1, rows_pad, Hd, Hd, ws + w.T2, rows_pad, rows_pad, sw)) return rc;
    // weight gradients reduce over the rows: few output tiles, long K -> split K so that the grid fills the chip
    auto wgrad_gemm = [&](const float *At, const float *amax_at, const float *Bt, const float *amax_bt, int Mo, int No, float *dst) -> int {
        const int tiles = (Mo / SplitGeom::BM) * (No / SplitGeom::BN);
        int split = 1;
        while (split < 16 && tiles * split < 256 && (rows_pad / GEMM_BK) % (2 * split) == 0) split *= 2;
        if (int rc = launch_gemm32<ACT_NONE, 4>(At, (int)rows_pad, amax_at, Bt, (int)rows_pad, amax_bt, nullptr, ws + w.splitk, No, nullptr, Mo, No,
                                                (int)rows_pad, split, sw))
            return rc;
        const size_t n = (size_t)Mo * No;
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, sw, ws + w.splitk, split, n, n, dst, 1.0f);
        return hipGetLastError() == hipSuccess ? 0 : fail(DNAG_E_HIP, "reduce_parts_kernel launch failed");
    };
    if (int rc = wgrad_gemm(ws + w.T1, amax_dh2, ws + w.T2, amax_h1, Hd, Hd, G + sl.f1_w)) return rc;   // dW2 = dH2^T H1
    // dH1 = (dH2 W1) * elu'(H1): the product, then the factor in a pass of its own (dnag_gemm16.hip.h mul_elugrad_kernel)
    if (int rc = launch_gemm32<ACT_NONE, 3>(ws + w.dh2, Hd, amax_dh2, P + tl.f1T, Hd, amax_w1, nullptr, ws + w.dh1, Hd, nullptr, rows_pad, Hd, Hd, 1, st)) return rc;
    hipLaunchKernelGGL(mul_elugrad_kernel, dim3(2048), dim3(256), 0, st, ws + w.dh1, ws + w.h1, rows_pad * (size_t)Hd / 4, amax_dh1);
    LAUNCH_CHECK();
    if (int rc = fork()) return rc;            // dH1 and its scale are complete
    if (int rc = launch_colsum(ws + w.dh1, Hd, rows, w.nz_rows, Hd, scr, G + sl.f0_b, sw)) return rc;
    if (side) HIP_TRY(hipEventRecord(side->mid, side->st));            // the second stream's last use of scr
    if (int rc = launch_transpose(ws + w.dh1, rows_pad, Hd, Hd, ws + w.T1, rows_pad, rows_pad, sw)) return rc;
    if (int rc = launch_transpose(ws + w.concat, rows_pad, F, F, ws + w.T2, rows_pad, rows_pad, sw)) return rc;
    if (int rc = wgrad_gemm(ws + w.T1, amax_dh1, ws + w.T2, amax_cc, Hd, F, G + sl.f0_w)) return rc;     // dW1 = dH1^T concat
    if (int rc = launch_gemm32<ACT_NONE, 3>(ws + w.dh1, Hd, amax_dh1, P + tl.f0T, Hd, amax_w0, nullptr, ws + w.dconcat, F, nullptr, rows_pad, F, Hd, 1, st)) return rc;
    // ---- 5. encoder backward -----------------------------------------------------------------------------------------
    if (int rc = train_g(B) == 8 ? launch_train_enc_bwd<8>(w.nblk_enc, P, tl, sv, ws + w.xt, ws + w.dconcat, p->mask, p->mask_stride ? 1 : 0, dm, ws + w.part, gl, ws + w.dtau, B, st)
                                 : launch_train_enc_bwd<4>(w.nblk_enc, P, tl, sv, ws + w.xt, ws + w.dconcat, p->mask, p->mask_stride ? 1 : 0, dm, ws + w.part, gl, ws + w.dtau, B, st))
        return rc;
    // ---- 6. time-embedding MLP backward ---------------------------------------------------------------------------------
    if (side) HIP_TRY(hipStreamWaitEvent(st, side->mid, 0));
    hipLaunchKernelGGL(time_embed_bwd_kernel, dim3(B), dim3(256), (size_t)d * sizeof(float), st, ws + w.dtau, ws + w.save_pre, d, P + e.t2_w, ws + w.dpre);
    LAUNCH_CHECK();
    if (int rc = launch_tn_small<1>(ws + w.dtau, d, ws + w.save_pre, 4 * d, B, w.nz_b, d, 4 * d, scr, G + sl.t2_w, st)) return rc;   // dW2 = dtau^T silu(pre)
    if (int rc = launch_colsum(ws + w.dtau, d, B, w.nz_b, d, scr, G + sl.t2_b, st)) return rc;
    if (int rc = launch_tn_small<0>(ws + w.dpre, 4 * d, ws + w.save_e, d, B, w.nz_b, 4 * d, d, scr, G + sl.t1_w, st)) return rc;    // dW1 = dpre^T e
    if (int rc = launch_colsum(ws + w.dpre, 4 * d, B, w.nz_b, 4 * d, scr, G + sl.t1_b, st)) return rc;
    // ---- 7. the encoder's weight gradients: slabs -> sums -> state order ---------------------------------------------------
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((gl.total + 255) / 256)), dim3(256), 0, st, ws + w.part, w.nblk_enc, gl.total, gl.total,
                       ws + w.gsum, 1.0f);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(train_unfold_layers_kernel, dim3(32, L), dim3(256), 0, st, ws + w.gsum, gl, S, sl, G, d, dff);
    LAUNCH_CHECK();
    {
        CopySegs gs;
        gs.count = 2;
        gs.src[0] = ws + w.gsum + gl.emb_w; gs.dst[0] = G + sl.emb_w; gs.n[0] = (size_t)d * V;
        gs.src[1] = ws + w.gsum + gl.emb_b; gs.dst[1] = G + sl.emb_b; gs.n[1] = (size_t)d;
        hipLaunchKernelGGL(copy_segs_kernel, dim3(2), dim3(256), 0, st, gs);
        LAUNCH_CHECK();
    }
    return join();
}

extern "C" size_t dnag_optim_workspace_bytes(void) { return OPTIM_BLOCKS * sizeof(double); }
extern "C" size_t dnag_optim_args_bytes(void) { return sizeof(dnag_optim_args); }

extern "C" int dnag_optim_step(const dnag_optim_args *p) {
    if (!p) return fail(DNAG_E_ARG, "args is NULL");
    if (p->n == 0) return 0;
    const bool adam = p->flags & DNAG_OPTIM_ADAM, ema = p->flags & DNAG_OPTIM_EMA;
    if (!adam && !ema) return fail(DNAG_E_ARG, "flags=%d selects neither DNAG_OPTIM_ADAM nor DNAG_OPTIM_EMA", p->flags);
    if (!p->params || (adam && (!p->grads || !p->exp_avg || !p->exp_avg_sq)) || (ema && !p->shadow)) return fail(DNAG_E_ARG, "NULL pointer");
    if (adam && (!(p->bc2_sqrt > 0.f) || !(p->eps >= 0.f) || !std::isfinite(p->step_size)))
        return fail(DNAG_E_ARG, "bc2_sqrt=%g eps=%g step_size=%g", p->bc2_sqrt, p->eps, p->step_size);
    hipStream_t st = (hipStream_t)p->stream;
    const bool need_norm = adam && (p->grad_clip >= 0.f || p->grad_norm_out);
    if (need_norm) {
        if (!p->workspace || p->workspace_bytes < dnag_optim_workspace_bytes())
            return fail(DNAG_E_WORKSPACE, "workspace %zu bytes < %zu needed", p->workspace_bytes, dnag_optim_workspace_bytes());
        hipLaunchKernelGGL(optim_sumsq_kernel, dim3(OPTIM_BLOCKS), dim3(256), 0, st, p->grads, p->n, (double *)p->workspace);
        LAUNCH_CHECK();
    }
    const size_t nb = (p->n + 255) / 256;
    hipLaunchKernelGGL(optim_update_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, st, *p, (const double *)p->workspace);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int dnag_nb201_seen_reset(int32_t *first_seen, void *stream) {
    if (!first_seen) return fail(DNAG_E_ARG, "NULL pointer");
    hipLaunchKernelGGL(fill_i32_kernel, dim3(16), dim3(1024), 0, (hipStream_t)stream, first_seen, (size_t)NB201_NCODES, (int32_t)NB201_UNSEEN);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int dnag_nb201_validity(const float *x, int batch, int32_t *status, int32_t *code, int32_t *first_seen, int32_t index_base,
                                   int32_t *counts, void *stream) {
    if (batch > 0 && (!x || !status || !code)) return fail(DNAG_E_ARG, "NULL pointer");
    if (batch < 0 || index_base < 0 || (int64_t)index_base + batch >= (int64_t)NB201_UNSEEN) return fail(DNAG_E_ARG, "batch=%d index_base=%d", batch, index_base);
    hipStream_t st = (hipStream_t)stream;
    if (batch > 0) {
        hipLaunchKernelGGL(nb201_validity_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, x, batch, status, code, first_seen, index_base);
        LAUNCH_CHECK();
    }
    if (counts) {
        hipLaunchKernelGGL(nb201_count_kernel, dim3(1), dim3(1024), 0, st, status, batch, first_seen, counts);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dnag_ofa_validity(const float *x, int batch, int32_t *status, int8_t *types, int32_t *counts, void *stream) {
    if (batch > 0 && (!x || !status || !types)) return fail(DNAG_E_ARG, "NULL pointer");
    if (batch < 0) return fail(DNAG_E_ARG, "batch=%d", batch);
    hipStream_t st = (hipStream_t)stream;
    if (batch > 0) {
        hipLaunchKernelGGL(ofa_validity_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, x, batch, status, types);
        LAUNCH_CHECK();
    }
    if (counts) {
        hipLaunchKernelGGL(ofa_count_kernel, dim3(1), dim3(1024), 0, st, status, batch, counts);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dnag_quantize(const float *x, float *q, size_t n, void *stream) {
    if (!x || !q) return fail(DNAG_E_ARG, "NULL pointer");
    if (n == 0) return 0;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(quantize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, q, n);
    LAUNCH_CHECK();
    return 0;
}
