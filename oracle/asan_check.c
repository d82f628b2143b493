/* oracle/asan_check.c -- TEST INFRASTRUCTURE.  Drives the CPU oracle under
 * AddressSanitizer/UBSan (the GPU pool has no sanitizer support; the reference itself
 * trips ASan at bipartite.h:65, which the restatement deliberately omits).
 * usage: asan_check <indptr.bin> <indices.bin> <seeds.bin> <batch> <n_parts> <f0,f1,..>
 * prints one checksum line over every exported list of strict and graph mode. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc orc;
orc* orc_create(const int64_t*, const int64_t*, int64_t, const int32_t*, int, int, const int32_t*, uint32_t, int);
void orc_destroy(orc*);
int orc_sample(orc*, const int64_t*, int64_t);
int orc_sample_graph(orc*, const int64_t*, int64_t);
int64_t orc_list_len(const orc*, int, int, int, int);
const int64_t* orc_list_ptr(const orc*, int, int, int, int);

static int64_t* slurp(const char* path, int64_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  int64_t* p = (int64_t*)malloc(sz ? sz : 8);
  if (fread(p, 1, sz, f) != (size_t)sz) exit(2);
  fclose(f);
  *n = sz / 8;
  return p;
}

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  int64_t n_ip, n_ix, n_seeds;
  int64_t* indptr = slurp(argv[1], &n_ip);
  int64_t* indices = slurp(argv[2], &n_ix);
  int64_t* seeds = slurp(argv[3], &n_seeds);
  int64_t batch = atoll(argv[4]);
  int P = atoi(argv[5]);
  int32_t fan[8];
  int L = 0;
  for (char* t = strtok(argv[6], ","); t && L < 8; t = strtok(NULL, ",")) fan[L++] = atoi(t);
  orc* o = orc_create(indptr, indices, n_ip - 1, NULL, P, L, fan, 5489u, 1);
  uint64_t sum = 1469598103934665603ull;
  for (int64_t b = 0; b * batch < n_seeds; b++) {
    int64_t n = n_seeds - b * batch < batch ? n_seeds - b * batch : batch;
    if (b & 1) orc_sample_graph(o, seeds + b * batch, n); else orc_sample(o, seeds + b * batch, n);
    for (int l = 0; l < L; l++)
      for (int g = 0; g < P; g++)
        for (int which = (b & 1) ? 100 : 0; which <= ((b & 1) ? 109 : 8); which++)
          for (int sub = 0; sub < P; sub++) {
            int64_t len = orc_list_len(o, l, g, which, sub);
            const int64_t* p = orc_list_ptr(o, l, g, which, sub);
            for (int64_t k = 0; k < len; k++) sum = (sum ^ (uint64_t)p[k]) * 1099511628211ull;
            sum = (sum ^ (uint64_t)len) * 1099511628211ull;
          }
  }
  orc_destroy(o);
  free(indptr); free(indices); free(seeds);
  printf("checksum %016llx\n", (unsigned long long)sum);
  return 0;
}
