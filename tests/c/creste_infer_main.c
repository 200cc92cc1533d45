/* A caller of libcreste_hip.so that is NOT Python: loads a plan exported by creste_public_amd.deploy.export_plan,
 * uploads two raw fp32 input files, runs creste_hip_model_infer and writes every output's bytes to <outdir>/<name>.bin
 * (the span the shape/strides cover).  tests/test_deploy_plan_gpu.py builds it with hipcc on the GPU box and compares
 * the files with the Python host path bit for bit.
 *   creste_infer_main <plan> <rgbd.f32> <p2p.f32> <outdir> [graph]                                              */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/creste_hip.h"

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(*n);
  if (fread(p, 1, *n, f) != *n) { fprintf(stderr, "short read %s\n", path); exit(2); }
  fclose(f);
  return p;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s plan rgbd.f32 p2p.f32 outdir [graph]\n", argv[0]); return 2; }
  void* h = NULL;
  if (creste_hip_model_load(argv[1], argc > 5 ? 1 : 0, &h) != 0) { fprintf(stderr, "load: %s\n", creste_last_error()); return 1; }
  printf("plan: %s\n", creste_hip_model_info(h));
  const int nin = creste_hip_model_num_inputs(h), nout = creste_hip_model_num_outputs(h);
  if (nin != 2) { fprintf(stderr, "expected 2 inputs\n"); return 1; }
  const void* dev_in[2];
  for (int i = 0; i < 2; ++i) {
    size_t n = 0;
    void* host = slurp(argv[2 + i], &n);
    void* d = NULL;
    if (hipMalloc(&d, n) != hipSuccess || hipMemcpy(d, host, n, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); return 1; }
    dev_in[i] = d;
    free(host);
  }
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return 1;
  for (int rep = 0; rep < 2; ++rep)       /* twice: the second run must not depend on state left by the first */
    if (creste_hip_model_infer(h, dev_in, 2, (void*)s) != 0) { fprintf(stderr, "infer: %s\n", creste_last_error()); return 1; }
  if (hipStreamSynchronize(s) != hipSuccess) { fprintf(stderr, "sync failed\n"); return 1; }
  for (int i = 0; i < nout; ++i) {
    const char* name; void* ptr; int dtype, ndim; int64_t shape[6], stride[6];
    if (creste_hip_model_output(h, i, &name, &ptr, &dtype, &ndim, shape, stride) != 0) return 1;
    int64_t span = 1;
    for (int k = 0; k < ndim; ++k) { if (shape[k] == 0) span = 0; }
    if (span) { span = 1; for (int k = 0; k < ndim; ++k) span += (shape[k] - 1) * stride[k]; }
    const size_t es = dtype == 0 ? 4 : dtype == 1 ? 8 : 1;
    void* host = malloc(span * es + 1);
    if (span && creste_hip_memcpy_d2h(host, ptr, span * es) != 0) { fprintf(stderr, "d2h: %s\n", creste_last_error()); return 1; }
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s.bin", argv[4], name);
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); return 1; }
    fwrite(host, es, span, f);
    fclose(f);
    free(host);
  }
  printf("wrote %d outputs\n", nout);
  creste_hip_model_free(h);
  return 0;
}
