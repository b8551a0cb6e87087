// Development probe: time the decode-step kernels in isolation (hipGraph of 200 dependent launches each).
#define AC_DEC_STAMPS 1
#include "../audiocaption_amd/csrc/gemm.hip"
#include "../audiocaption_amd/csrc/decoder.hip"
#include <stdio.h>
#include <vector>

template <typename F>
void run(const char* name, F launch, hipStream_t s) {
  for (int i = 0; i < 5; ++i) launch();
  (void)hipStreamSynchronize(s);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 200; ++i) launch();
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a, s);
  for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-44s %6.2f us/kernel\n", name, ms * 1000 / 2000);
}

int main() {
  hipStream_t s; (void)hipStreamCreate(&s);
  const int R = 64, d = 256, ff = 1024, V = 4368;
  float *x, *y, *w, *wp, *out, *lnw, *lnb, *big;
  (void)hipMalloc(&x, R * ff * 4); (void)hipMalloc(&y, R * ff * 4); (void)hipMalloc(&w, (size_t)4400 * ff * 4);
  (void)hipMalloc(&wp, (size_t)4400 * ff * 4); (void)hipMalloc(&out, (size_t)R * 4400 * 4);
  (void)hipMalloc(&lnw, ff * 4); (void)hipMalloc(&lnb, ff * 4); (void)hipMalloc(&big, (size_t)R * 20 * d * 4 * 2);
  (void)hipMemset(x, 0, R * ff * 4); (void)hipMemset(y, 0, R * ff * 4); (void)hipMemset(w, 0, (size_t)4400 * ff * 4);
  (void)hipMemset(wp, 0, (size_t)4400 * ff * 4); (void)hipMemset(lnw, 0, ff * 4); (void)hipMemset(lnb, 0, ff * 4);
  DecGemmParams g;
  g.tok = nullptr; g.tok_stride = 0; g.t = 0; g.emb = nullptr; g.pe = nullptr; g.emb_scale = 1.f;
  g.M = R; g.X = x; g.ldx = d; g.Y2 = y; g.ldy2 = d; g.ln_w = lnw; g.ln_b = lnb; g.xout = big; g.ldxo = d;
  g.Wp = wp; g.bias = lnb; g.Y = out; g.relu = 0; g.ntb = 1;
  auto G = [&](int pro, int N, int K, int ldx) {
    g.N = N; g.K = K; g.ldx = ldx; g.ldy = N;
    if (pro == 0) launch_dec_gemm<PRO_PLAIN>(g, s); else launch_dec_gemm<PRO_ADDLN>(g, s);
  };
  run("dec_gemm PLAIN N=256 K=256", [&] { G(0, 256, 256, 256); }, s);
  run("dec_gemm PLAIN N=256 K=1024", [&] { G(0, 256, 1024, 1024); }, s);
  run("dec_gemm ADDLN N=256 K=256", [&] { G(2, 256, 256, 256); }, s);
  run("dec_gemm ADDLN N=768 K=256", [&] { G(2, 768, 256, 256); }, s);
  run("dec_gemm ADDLN N=1024 K=256", [&] { G(2, 1024, 256, 256); }, s);
  run("dec_gemm ADDLN N=4368 K=256", [&] { G(2, 4368, 256, 256); }, s);
  g.ntb = 4; run("dec_gemm ADDLN N=4368 K=256 ntb=4", [&] { G(2, 4368, 256, 256); }, s);
  g.ntb = 8; run("dec_gemm ADDLN N=4368 K=256 ntb=8", [&] { G(2, 4368, 256, 256); }, s); g.ntb = 1;
  run("add_layernorm 64x256", [&] { launch_ln(x, y, lnw, lnb, out, R, d, d, d, d, s); }, s);
  run("ac_linear(skinny) 64x256x256", [&] { ac_linear(x, w, lnb, out, R, 256, 256, 256, 256, 256, 0, s); }, s);
  run("ac_linear(skinny) 64x256x1024", [&] { ac_linear(x, w, lnb, out, R, 256, 1024, 1024, 1024, 256, 0, s); }, s);
  run("ac_linear(skinny) 64x4368x256", [&] { ac_linear(x, w, lnb, out, R, 4368, 256, 256, 256, 4368, 0, s); }, s);
  // phase stamps of one launch of each flavour
  auto stamps = [&](const char* name) {
    (void)hipStreamSynchronize(s);
    long long h[16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dec_stamps), sizeof(h));
    printf("%-28s cycles@2.4GHz: b-load-issue..producer-loads %lld | LN %lld | sync+stage %lld | mfma %lld | reduce+store %lld | total %lld (%.2f us)\n",
           name, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[5] - h[0], (h[5] - h[0]) / 2400.0);
  };
  G(2, 256, 256, 256); stamps("ADDLN N=256 K=256");
  G(2, 4368, 256, 256); stamps("ADDLN N=4368 K=256");
  G(0, 256, 256, 256); stamps("PLAIN N=256 K=256");
  G(0, 256, 1024, 1024); stamps("PLAIN N=256 K=1024");
  return 0;
}
