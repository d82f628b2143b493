// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin driver around the UNMODIFIED reference cslicer sources.  It is compiled
// by oracle/Makefile against the sources where they lie under
// /root/reference/cslicer (never copied into this repo); the binary lands in
// oracle/_ref/ (git-ignored).  It exists to
//   (1) dump golden vectors: every list of the 3x4 BiPartite objects the
//       reference hands to Python (PySample, pybipartite.cpp:49-66), plus the
//       raw per-node output of Slicer::neighbour_sample (slicer.cpp:6-22)
//       captured with a second, identically seeded Slicer;
//   (2) time the reference slicer (cpu_baseline kind "reference") at its
//       hard-coded constants: fanout 10/10/10, 4 parts, workload v%4.
//
// The reference worker loop Slicer::run (slicer.cpp:120-140) is driven exactly
// as WorkerPool does (WorkerPool.cpp:29-33,41-55): batches pushed into the work
// queue followed by one empty sentinel vector, PySamples popped from the
// generated-samples queue.
//
// slicer.cpp is included into this TU because neighbour_sample is declared
// `inline` inside the .cpp (slicer.cpp:6) and is otherwise not linkable.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "slicer.cpp"  // resolved by -I /root/reference/cslicer

namespace {

typedef std::vector<long> vl;

struct NullBuf : std::streambuf {
  int overflow(int c) override { return c; }
};

void put(FILE* f, long v) { fwrite(&v, sizeof(long), 1, f); }
void put_list(FILE* f, const vl& v) {
  put(f, (long)v.size());
  if (!v.empty()) fwrite(v.data(), sizeof(long), v.size(), f);
}

// batches file: int64 words: n_batches, then per batch: n, ids...
std::vector<vl> read_batches(const char* path) {
  std::vector<vl> out;
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  long nb = 0;
  if (fread(&nb, sizeof(long), 1, f) != 1) exit(2);
  for (long b = 0; b < nb; b++) {
    long n = 0;
    if (fread(&n, sizeof(long), 1, f) != 1) exit(2);
    vl v(n);
    if (n && fread(v.data(), sizeof(long), n, f) != (size_t)n) exit(2);
    out.push_back(v);
  }
  fclose(f);
  return out;
}

struct World {
  Dataset* dataset;
  std::vector<int> workload;
  std::vector<int>* storage[4];
  explicit World(const std::string& dir) {
    dataset = new Dataset(dir);
    // pyfrontend.cpp:53-65: workload_map[j] = j % 4, storage_map[k][j] = (k == j%4)
    for (int k = 0; k < 4; k++) storage[k] = new std::vector<int>();
    for (long j = 0; j < dataset->num_nodes; j++) {
      workload.push_back(j % 4);
      for (int k = 0; k < 4; k++) storage[k]->push_back(k == j % 4 ? 1 : 0);
    }
  }
};

int do_dump(const char* dir, const char* batches_path, const char* out_path) {
  World w(dir);
  std::vector<vl> batches = read_batches(batches_path);
  ConQueue<PySample*> gen((int)batches.size() + 4);
  ConQueue<std::vector<long>*> work((int)batches.size() + 4);
  // A: the slicer whose output is the golden object.  B: identically seeded
  // twin used only to replay neighbour_sample in the same order.
  Slicer A(w.dataset, &w.workload, w.storage, 0, &gen, &work);
  Slicer B(w.dataset, &w.workload, w.storage, 0, &gen, &work);
  for (auto& b : batches) work.push_object(new vl(b));
  work.push_object(new vl());  // end-of-work sentinel, WorkerPool.cpp:52-55
  A.run();

  FILE* f = fopen(out_path, "wb");
  if (!f) { fprintf(stderr, "cannot open %s\n", out_path); return 2; }
  put(f, 0x43534C4F52433031L);  // magic
  put(f, (long)batches.size());
  put(f, w.dataset->num_nodes);
  std::vector<int> seen(w.dataset->num_nodes, 0);
  for (size_t bi = 0; bi < batches.size(); bi++) {
    PySample* s = gen.pop_object();
    put_list(f, batches[bi]);
    vl in = batches[bi];
    for (int l = 0; l < 3; l++) {
      // replay of the traversal order of slice_layer (slicer.cpp:26-50)
      vl flat;      // concatenated neighbour_sample outputs
      vl counts;    // entries per frontier node (incl. the leading self entry)
      vl next;
      long draws = 0;
      for (long nd1 : in) {
        vl nb;
        B.neighbour_sample(nd1, nb);
        long deg = w.dataset->indptr[nd1 + 1] - w.dataset->indptr[nd1];
        if (deg >= 10) draws += 10;
        counts.push_back((long)nb.size());
        for (long nd2 : nb) {
          flat.push_back(nd2);
          if (!seen[nd2]) { seen[nd2] = 1; next.push_back(nd2); }
        }
      }
      for (long v : next) seen[v] = 0;
      put_list(f, in);
      put_list(f, counts);
      put_list(f, flat);
      put_list(f, next);
      put(f, draws);
      for (int g = 0; g < 4; g++) {
        PyBipartite* bp = (*s->layers[l])[g];
        put(f, (long)bp->gpu_id);
        put_list(f, bp->in_nodes);
        put_list(f, bp->indptr);
        put_list(f, bp->out_nodes);
        put_list(f, bp->owned_out_nodes);
        put_list(f, bp->indices);
        put_list(f, bp->self_ids_in);
        put_list(f, bp->self_ids_out);
        for (int k = 0; k < 4; k++) put_list(f, bp->from_ids[k]);
        for (int k = 0; k < 4; k++) put_list(f, bp->to_ids[k]);
      }
      in = next;
    }
    delete s;
  }
  fclose(f);
  return 0;
}

// Timing: T Slicers, one thread each (WorkerPool.cpp:29-33); batches dealt
// round-robin (batch b -> worker b % T, as the stale cslicer/driver.cpp:69-71
// does).  Each worker gets its own queues so the assignment is deterministic.
// Time = first pop to last push, includes the PySample deep copy.
int do_bench(const char* dir, const char* batches_path, int T) {
  World w(dir);
  std::vector<vl> batches = read_batches(batches_path);
  std::vector<ConQueue<PySample*>*> gen(T);
  std::vector<ConQueue<std::vector<long>*>*> work(T);
  std::vector<Slicer*> sl(T);
  for (int t = 0; t < T; t++) {
    gen[t] = new ConQueue<PySample*>((int)batches.size() + 4);
    work[t] = new ConQueue<std::vector<long>*>((int)batches.size() + 4);
    sl[t] = new Slicer(w.dataset, &w.workload, w.storage, 0, gen[t], work[t]);
  }
  for (size_t b = 0; b < batches.size(); b++) work[b % T]->push_object(new vl(batches[b]));
  for (int t = 0; t < T; t++) work[t]->push_object(new vl());
  std::vector<std::thread> th;
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int t = 0; t < T; t++) th.emplace_back(&Slicer::run, sl[t]);
  for (auto& x : th) x.join();
  auto t1 = std::chrono::high_resolution_clock::now();
  double sec = std::chrono::duration<double>(t1 - t0).count();
  fprintf(stderr, "REF_BENCH threads=%d batches=%zu seconds=%.6f iters_per_sec=%.4f\n", T,
          batches.size(), sec, batches.size() / sec);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  // the reference prints on every push/pop (slicer.cpp:123,132,137; conqueue.h:44)
  static NullBuf nb;
  std::cout.rdbuf(&nb);
  if (argc >= 5 && std::string(argv[1]) == "dump") return do_dump(argv[2], argv[3], argv[4]);
  if (argc >= 5 && std::string(argv[1]) == "bench") return do_bench(argv[2], argv[3], atoi(argv[4]));
  fprintf(stderr,
          "usage: ref_harness dump <l0_dir> <batches.bin> <out.bin>\n"
          "       ref_harness bench <l0_dir> <batches.bin> <threads>\n");
  return 2;
}
