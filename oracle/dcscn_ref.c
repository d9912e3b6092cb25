/*
 * dcscn_ref.c -- plain-C restatement of the DCSCN layer semantics.  TEST INFRASTRUCTURE ONLY: it is
 * compiled into oracle/libdcscn_ref.so by oracle/Makefile and called from tests (through ctypes) as a
 * third, independent implementation next to oracle/dcscn_oracle.py (numpy) and oracle/cpu_path_torch.py.
 * Nothing in the shipped package links or loads it.
 *
 * Each function restates one TensorFlow op as the reference uses it (the arithmetic of the reference
 * lives in TensorFlow, un-pinned `tensorflow>=2.0.0`, Pipfile:9):
 *   ref_conv2d_same      tf.nn.conv2d, stride 1, SAME, NHWC x HWIO            helper/tf_graph.py:105
 *   ref_depthwise_same   tf.nn.depthwise_conv2d, channel multiplier 1          helper/tf_graph.py:161
 *   ref_bias_act         tf.add(bias) + build_activator                        helper/tf_graph.py:77-102,109
 *   ref_depth_to_space   tf.depth_to_space                                     helper/tf_graph.py:248
 * All arithmetic in double; tensors are dense NHWC double arrays.  Parity pinning: see the header of
 * dcscn_oracle.py (README PSNR table; bit-level parity with the reference binary is unpinned).
 */
#include <math.h>
#include <stddef.h>

void ref_conv2d_same(const double* x, const double* w, double* y, int n, int h, int wd, int cin, int cout, int k) {
    const int pad = k / 2;
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < wd; ++j) {
                double* out = y + (((size_t)b * h + i) * wd + j) * cout;
                for (int o = 0; o < cout; ++o) out[o] = 0.0;
                for (int dy = 0; dy < k; ++dy) {
                    const int yy = i + dy - pad;
                    if (yy < 0 || yy >= h) continue;          /* SAME: zeros outside the image */
                    for (int dx = 0; dx < k; ++dx) {
                        const int xx = j + dx - pad;
                        if (xx < 0 || xx >= wd) continue;
                        const double* in = x + (((size_t)b * h + yy) * wd + xx) * cin;
                        const double* wt = w + ((size_t)(dy * k + dx) * cin) * cout;
                        for (int c = 0; c < cin; ++c) {
                            const double v = in[c];
                            const double* wr = wt + (size_t)c * cout;
                            for (int o = 0; o < cout; ++o) out[o] += v * wr[o];
                        }
                    }
                }
            }
}

void ref_depthwise_same(const double* x, const double* w, double* y, int n, int h, int wd, int c, int k) {
    const int pad = k / 2;
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < wd; ++j) {
                double* out = y + (((size_t)b * h + i) * wd + j) * c;
                for (int ch = 0; ch < c; ++ch) out[ch] = 0.0;
                for (int dy = 0; dy < k; ++dy) {
                    const int yy = i + dy - pad;
                    if (yy < 0 || yy >= h) continue;
                    for (int dx = 0; dx < k; ++dx) {
                        const int xx = j + dx - pad;
                        if (xx < 0 || xx >= wd) continue;
                        const double* in = x + (((size_t)b * h + yy) * wd + xx) * c;
                        const double* wt = w + (size_t)(dy * k + dx) * c;      /* [k,k,C,1] */
                        for (int ch = 0; ch < c; ++ch) out[ch] += in[ch] * wt[ch];
                    }
                }
            }
}

/* act: 0 none, 1 prelu (alpha per channel), 2 relu, 3 leaky_relu(0.1), 4 sigmoid, 5 tanh, 6 selu */
void ref_bias_act(double* y, const double* bias, const double* alpha, size_t pixels, int c, int act) {
    for (size_t p = 0; p < pixels; ++p) {
        double* v = y + p * c;
        for (int ch = 0; ch < c; ++ch) {
            double t = v[ch] + (bias ? bias[ch] : 0.0);
            switch (act) {
                case 1: t = (t > 0 ? t : 0.0) + alpha[ch] * (t - fabs(t)) * 0.5; break;   /* tf_graph.py:94 */
                case 2: t = t > 0 ? t : 0.0; break;
                case 3: t = t > 0.1 * t ? t : 0.1 * t; break;
                case 4: t = 1.0 / (1.0 + exp(-t)); break;
                case 5: t = tanh(t); break;
                case 6: t = 1.0507009873554804934193349852946 * (t > 0 ? t : 1.6732632423543772848170429916717 * (exp(t) - 1.0)); break;
                default: break;
            }
            v[ch] = t;
        }
    }
}

void ref_depth_to_space(const double* x, double* y, int n, int h, int wd, int c, int block) {
    const int co = c / (block * block);
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < wd; ++j)
                for (int bi = 0; bi < block; ++bi)
                    for (int bj = 0; bj < block; ++bj)
                        for (int ch = 0; ch < co; ++ch)
                            y[(((size_t)b * h * block + (size_t)i * block + bi) * ((size_t)wd * block) + (size_t)j * block + bj) * co + ch] =
                                x[(((size_t)b * h + i) * wd + j) * c + (bi * block + bj) * co + ch];
}
