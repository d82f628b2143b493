// pymodule.cpp -- native host side of the drop-in: the pybind11 module `cslicer`
// with the reference module's exact surface (cslicer/pyfrontend.cpp:116-148),
// over the C ABI of the HIP engine (include/cslicer_hip.h).
//
//   reference                         here
//   Dataset (dataset.cpp:8-113)       L0Dataset: same files, checksums enforced
//   WorkerPool::run (WorkerPool.cpp)  Producer thread: epoch shuffle with the same
//                                     std::random_shuffle call, rounds of S
//                                     minibatches submitted to the GPU
//   Slicer workers (slicer.cpp)       engine streams (one mt19937(5489) each)
//   ConQueue (util/conqueue.h)        ReadyQueue: bounded, blocking, GIL released
//   PySample/PyBipartite              same structs, same def_readwrite members
//
// The producer keeps two rounds in flight (two result slots): while the GPU
// slices round r+1 the host copies round r's lists into PySample objects.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cslicer_hip.h"

namespace py = pybind11;

namespace {

const char* kDefaultRoot = "/data/sandeep/";  // pyfrontend.cpp:24

// ---- PyBipartite / PySample: pybipartite.h:10-47, same members, same types
struct PyBipartite {
  std::vector<long> in_nodes, indptr, out_nodes, owned_out_nodes, indices;
  std::vector<std::vector<long>> from_ids, to_ids;
  std::vector<long> self_ids_in, self_ids_out;
  int gpu_id = -1;
};

// A sample the consumer lets go of hands its BiPartites (a dozen objects, ~100 vectors, megabytes of `long`) to this
// recycler instead of freeing them on the consumer's thread -- "the consumer's own deletes of the previous sample set the
// pace" of the drop-in path in round 2 -- and the converter fills recycled ones: their vectors keep their capacity, so a
// steady state neither allocates nor frees nor page-faults.  Bounded (the queue holds 10 samples, a few more are in flight);
// what does not fit is freed as before.  CSLICER_NO_RECYCLE=1: A/B switch.
class Recycler {
 public:
  typedef std::vector<std::vector<PyBipartite*>*> Layers;
  static Recycler& get() {
    static Recycler* r = new Recycler();   // (never destroyed: a sample may outlive every static object at interpreter exit)
    return *r;
  }
  // true: taken over (the caller must not free them)
  bool give(Layers& layers) {
    if (off_ || layers.empty()) return false;
    std::lock_guard<std::mutex> lk(m_);
    if (free_.size() >= kMax) return false;
    free_.emplace_back(std::move(layers));
    layers.clear();
    return true;
  }
  bool take(int n_layers, int n_parts, Layers* out) {
    if (off_) return false;
    std::lock_guard<std::mutex> lk(m_);
    for (size_t i = free_.size(); i-- > 0;) {
      Layers& c = free_[i];
      if ((int)c.size() == n_layers && (int)c[0]->size() == n_parts && (int)(*c[0])[0]->from_ids.size() == n_parts) {
        *out = std::move(c);
        free_.erase(free_.begin() + (long)i);
        return true;
      }
    }
    return false;
  }
  static void destroy(Layers& layers) {
    for (auto l : layers) {
      for (auto b : *l) delete b;
      delete l;
    }
    layers.clear();
  }
  ~Recycler() {
    for (auto& c : free_) destroy(c);
  }

 private:
  Recycler() : off_(getenv("CSLICER_NO_RECYCLE") != nullptr) {}
  static constexpr size_t kMax = 24;
  const bool off_;
  std::mutex m_;
  std::vector<Layers> free_;
};

struct PySample {
  std::vector<std::vector<PyBipartite*>*> layers;
  long in_nodes = 0, out_nodes = 0;  // pybipartite.cpp:57-62 (not exported there either)
  ~PySample() {
    if (!Recycler::get().give(layers)) Recycler::destroy(layers);
  }
};

PySample* empty_sample(int n_layers, int n_parts) {
  PySample* s = new PySample();
  if (Recycler::get().take(n_layers, n_parts, &s->layers)) {
    // recycled
    for (int l = 0; l < n_layers; l++) {
      for (int g = 0; g < n_parts; g++) {
        PyBipartite* b = (*s->layers[l])[g];
        b->gpu_id = g;
        // (clear() keeps the capacity; the lists are Python-writable, so nothing of the previous sample may show through)
        b->in_nodes.clear(), b->indptr.clear(), b->out_nodes.clear(), b->owned_out_nodes.clear(), b->indices.clear();
        b->self_ids_in.clear(), b->self_ids_out.clear();
        for (auto& v : b->from_ids) v.clear();
        for (auto& v : b->to_ids) v.clear();
      }
    }
    return s;
  }
  for (int l = 0; l < n_layers; l++) {
    auto row = new std::vector<PyBipartite*>();
    for (int g = 0; g < n_parts; g++) {
      auto b = new PyBipartite();
      b->gpu_id = g;
      b->from_ids.resize(n_parts);
      b->to_ids.resize(n_parts);
      row->push_back(b);
    }
    s->layers.push_back(row);
  }
  return s;
}

// ---- Dataset: dataset.cpp:8-113 (graph part; features/labels/partition map are
// loaded by the reference but never used by the slicer)
struct L0Dataset {
  long num_nodes = -1, num_edges = -1, csum_offsets = 0, csum_edges = 0;
  std::vector<long> indptr, indices;
  explicit L0Dataset(const std::string& dir) {
    std::ifstream meta(dir + "/meta.txt");
    if (!meta) throw std::runtime_error("cslicer: cannot open " + dir + "/meta.txt");
    std::string line;
    while (std::getline(meta, line)) {
      if (meta.eof()) break;  // dataset.cpp:75: an unterminated last line is dropped
      auto eq = line.find('=');
      if (eq == std::string::npos) continue;
      const std::string name = line.substr(0, eq);
      const long val = std::stoll(line.substr(eq + 1));
      if (name == "num_nodes") num_nodes = val;
      if (name == "num_edges") num_edges = val;
      if (name == "csum_offsets") csum_offsets = val;
      if (name == "csum_edges") csum_edges = val;
    }
    if (num_nodes < 1 || num_edges < 0) throw std::runtime_error("cslicer: meta.txt lacks num_nodes/num_edges");
    indptr.resize(num_nodes + 1);
    indices.resize(num_edges);
    read_all(dir + "/indptr.bin", indptr);
    read_all(dir + "/indices.bin", indices);
    long s = 0;
    for (long v : indptr) s += v;
    if (s != csum_offsets) throw std::runtime_error("cslicer: indptr checksum mismatch (dataset.cpp:27)");
    s = 0;
    for (long v : indices) s += v;
    if (s != csum_edges) throw std::runtime_error("cslicer: indices checksum mismatch (dataset.cpp:35)");
  }
  static void read_all(const std::string& path, std::vector<long>& dst) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cslicer: cannot open " + path);
    f.read(reinterpret_cast<char*>(dst.data()), (std::streamsize)(dst.size() * sizeof(long)));
    if ((size_t)f.gcount() != dst.size() * sizeof(long)) throw std::runtime_error("cslicer: short read of " + path);
  }
};

// A few helper threads for the int32 -> long deep copy of one sample (12 bipartites, the deepest layer's
// four hold ~80 % of the ids): the caller takes part, tasks are claimed with an atomic counter.
class MiniPool {
 public:
  explicit MiniPool(int n) {
    for (int i = 0; i < n; i++) th_.emplace_back([this] { worker(); });
  }
  ~MiniPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(int n_tasks, const std::function<void(int)>& fn) {
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      n_tasks_ = n_tasks;
      pending_ = n_tasks;
      gen_++;
      next_.store(0);  // last: a helper that sees the new counter sees the new task set
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
  }

 private:
  void work() {
    for (;;) {
      const int t = next_.fetch_add(1);
      if (t >= n_tasks_) return;
      (*fn_)(t);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_tasks_ = 0, pending_ = 0;
  std::atomic<int> next_{1 << 30};
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// ---- ConQueue<PySample*>: util/conqueue.h:10-63, plus a close() so that the
// destructor cannot deadlock (the reference's ~CSlicer joins a producer that may
// be blocked on a full queue, pyfrontend.cpp:85-88)
class ReadyQueue {
  size_t max_size_;
  std::queue<PySample*> q_;
  std::mutex m_;
  std::condition_variable has_space_, not_empty_;
  bool closed_ = false;
  std::string error_;

 public:
  explicit ReadyQueue(size_t n) : max_size_(n) {}
  bool push(PySample* s) {
    std::unique_lock<std::mutex> lk(m_);
    has_space_.wait(lk, [&] { return q_.size() < max_size_ || closed_; });
    if (closed_) return false;
    q_.push(s);
    not_empty_.notify_all();
    return true;
  }
  PySample* pop() {
    std::unique_lock<std::mutex> lk(m_);
    not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) throw std::runtime_error(error_.empty() ? "cslicer: queue closed" : error_);
    PySample* s = q_.front();
    q_.pop();
    has_space_.notify_all();
    return s;
  }
  void close(const std::string& err = "") {
    std::lock_guard<std::mutex> lk(m_);
    closed_ = true;
    if (!err.empty()) error_ = err;
    has_space_.notify_all();
    not_empty_.notify_all();
  }
  void drain() {
    std::lock_guard<std::mutex> lk(m_);
    while (!q_.empty()) {
      delete q_.front();
      q_.pop();
    }
  }
};

void check(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string("cslicer: ") + what + ": " + csl_last_error());
}

// ---- CSlicer: pyfrontend.cpp:25-89
class CSlicer {
  std::string name_;
  int queue_size_, workers_, epochs_, batch_;
  int n_parts_, n_layers_;
  long num_nodes_ = 0;
  L0Dataset* dataset_ = nullptr;
  csl_engine* eng_ = nullptr;
  ReadyQueue ready_;
  std::thread producer_;
  std::atomic<bool> stop_{false};
  std::atomic<long> handed_{0};
  bool shuffle_;

 public:
  CSlicer(const std::string& name, int queue_size, int no_worker_threads, int number_of_epochs, int minibatch_size,
          const std::string& data_root, const std::vector<int>& fanout, int n_parts, int device, unsigned seed,
          bool shuffle, const std::string& partition)
      : queue_size_(queue_size),
        workers_(std::max(1, no_worker_threads)),
        epochs_(number_of_epochs),
        batch_(minibatch_size),
        n_parts_(n_parts),
        n_layers_((int)fanout.size()),
        ready_(10),  // WorkerPool.cpp:23-24: capacity 10, the queue_size argument is ignored
        shuffle_(shuffle) {
    std::string root = data_root;
    if (root.empty()) {
      const char* env = std::getenv("CSLICER_DATA_ROOT");
      root = env ? env : kDefaultRoot;
    }
    if (!root.empty() && root.back() != '/') root += "/";
    name_ = root + name;
    if (n_layers_ < 1 || n_layers_ > CSL_MAX_LAYERS) throw std::runtime_error("cslicer: 1..4 layers");
    dataset_ = new L0Dataset(name_);
    num_nodes_ = dataset_->num_nodes;
    csl_config c;
    std::memset(&c, 0, sizeof(c));
    c.abi_version = CSL_ABI_VERSION;
    c.device = device;
    c.num_nodes = dataset_->num_nodes;
    c.num_edges = dataset_->num_edges;
    c.indptr = reinterpret_cast<const int64_t*>(dataset_->indptr.data());
    c.indices = reinterpret_cast<const int64_t*>(dataset_->indices.data());
    // the reference loads partition_map_opt.bin (dataset.cpp:59-67) but slices by v % 4
    // (pyfrontend.cpp:57); partition="file" uses the map (METIS output of python/utils/metis.py)
    std::vector<int> part_map;
    c.workload = nullptr;
    if (partition == "file") {
      part_map.resize(num_nodes_);
      std::ifstream f(name_ + "/partition_map_opt.bin", std::ios::binary);
      if (!f) throw std::runtime_error("cslicer: cannot open " + name_ + "/partition_map_opt.bin");
      f.read(reinterpret_cast<char*>(part_map.data()), (std::streamsize)(part_map.size() * sizeof(int)));
      if ((size_t)f.gcount() != part_map.size() * sizeof(int))
        throw std::runtime_error("cslicer: partition_map_opt.bin is shorter than num_nodes");
      c.workload = part_map.data();
    } else if (partition != "mod") {
      throw std::runtime_error("cslicer: partition must be 'mod' or 'file'");
    }
    c.n_parts = n_parts;
    c.n_layers = n_layers_;
    for (int l = 0; l < n_layers_; l++) c.fanout[l] = fanout[l];
    c.max_batch = minibatch_size;
    c.n_streams = workers_;
    c.n_slots = 2;
    c.rng_seed = seed;
    if (csl_create(&c, &eng_) < 0) {
      delete dataset_;
      dataset_ = nullptr;
      throw std::runtime_error(std::string("cslicer: csl_create: ") + csl_last_error());
    }
    // the CSR now lives on the GPU; the host copy is not needed any more
    std::vector<long>().swap(dataset_->indices);
    std::vector<long>().swap(dataset_->indptr);
    producer_ = std::thread(&CSlicer::run, this);  // pyfrontend.cpp:69
  }

  ~CSlicer() {
    stop_ = true;
    ready_.close();
    if (producer_.joinable()) producer_.join();
    ready_.drain();
    if (eng_) csl_destroy(eng_);
    delete dataset_;
  }

  long expected_number_of_samples() const {  // pyfrontend.cpp:80-83
    const long per_epoch = (num_nodes_ - 1) / batch_ + 1;
    return per_epoch * epochs_;
  }

  PySample* getSample() {  // pyfrontend.cpp:73-78 -> WorkerPool::pop_object
    if (handed_ >= expected_number_of_samples())
      throw std::runtime_error("cslicer: all samples already consumed (the reference would block forever)");
    PySample* s = ready_.pop();
    handed_++;
    return s;
  }

 private:
  // One sample on its way out: the packed int32 lists in the engine's pinned staging (two buffers alternate,
  // so the converter may still read sample k while the producer fetches sample k+1) + where each list sits.
  struct Staged {
    csl_sample_meta m;
    const int32_t* host = nullptr;
    int64_t seg[CSL_MAX_LAYERS][CSL_NUM_LISTS];
  };

  void fetch(int slot, int stream, Staged* st) {
    check(csl_fetch_sample32(eng_, slot, stream, &st->m, &st->host, st->seg), "csl_fetch_sample32");
  }

  // PySample(Sample*) (pybipartite.cpp:54-66): the deep copy into `long` vectors, int32 -> long on the way
  PySample* convert(const Staged& st) {
    const csl_sample_meta& m = st.m;
    PySample* s = empty_sample(n_layers_, n_parts_);
    auto take = [&](int l, int kind, int g, std::vector<long>& dst) {
      const long lo = (long)m.layer[l].off[kind][g], hi = (long)m.layer[l].off[kind][g + 1];
      const int32_t* src = st.host + st.seg[l][kind] + lo;
      dst.assign(src, src + (hi - lo));  // int32 -> long
    };
    const std::function<void(int)> one = [&](int t) {
      const int l = t / n_parts_, g = t % n_parts_;
      PyBipartite* b = (*s->layers[l])[g];
      take(l, CSL_IN_NODES, g, b->in_nodes);
      take(l, CSL_OUT_NODES, g, b->out_nodes);
      take(l, CSL_OWNED_OUT_NODES, g, b->owned_out_nodes);
      take(l, CSL_SELF_IDS_IN, g, b->self_ids_in);
      take(l, CSL_SELF_IDS_OUT, g, b->self_ids_out);
      take(l, CSL_TO_IDS, g, b->to_ids[g]);      // own index only, slicer.cpp:41
      take(l, CSL_FROM_IDS, g, b->from_ids[g]);  // slicer.cpp:42
      b->indptr.assign((size_t)m.layer[l].indptr_len[g], 1);  // bipartite.h:55-66: one `1` per push, CSR never built
    };
    // deepest layer first: its bipartites are the big tasks
    const std::function<void(int)> rev = [&](int t) { one(n_layers_ * n_parts_ - 1 - t); };
    pool_.run(n_layers_ * n_parts_, rev);
    for (int l = 0; l < n_layers_; l++) {
      for (int g = 0; g < n_parts_; g++) {
        PyBipartite* b = (*s->layers[l])[g];
        if (l == 0) s->in_nodes += (long)b->in_nodes.size();
        if (g == 2) s->out_nodes += (long)b->out_nodes.size();
      }
    }
    return s;
  }

  // Hand-over producer -> converter, one sample deep (the staging has two buffers): D2H of sample k+1
  // overlaps the widening of sample k.  Returns false when the consumer has gone away.
  bool hand_over(const Staged& st) {
    std::unique_lock<std::mutex> lk(cv_m_);
    cv_.wait(lk, [&] { return !have_staged_ || conv_stop_; });
    if (conv_stop_) return false;
    staged_ = st;
    have_staged_ = true;
    cv_.notify_all();
    return true;
  }

  void converter() {
    for (;;) {
      Staged st;
      {
        std::unique_lock<std::mutex> lk(cv_m_);
        cv_.wait(lk, [&] { return have_staged_ || conv_done_ || conv_stop_; });
        if (conv_stop_ || (!have_staged_ && conv_done_)) return;
        st = staged_;
      }
      PySample* smp = convert(st);   // reads st.host while the producer may already fetch the next sample
      {
        std::lock_guard<std::mutex> lk(cv_m_);
        have_staged_ = false;        // the staging buffer two fetches back is free again
        cv_.notify_all();
      }
      if (!ready_.push(smp)) {
        delete smp;
        std::lock_guard<std::mutex> lk(cv_m_);
        conv_stop_ = true;
        cv_.notify_all();
        return;
      }
    }
  }

  // WorkerPool::run (WorkerPool.cpp:37-60)
  void run() {
    std::thread conv(&CSlicer::converter, this);
    try {
      std::vector<long> nodes(num_nodes_);
      for (long i = 0; i < num_nodes_; i++) nodes[i] = i;  // WorkerPool.cpp:12-16
      const long per_epoch = (num_nodes_ - 1) / batch_ + 1;
      const long rounds = (per_epoch + workers_ - 1) / workers_;
      bool alive = true;
      for (int epoch = 0; epoch < epochs_ && !stop_ && alive; epoch++) {
        if (shuffle_) std::random_shuffle(nodes.begin(), nodes.end());  // WorkerPool.cpp:40
        check(csl_set_nodes(eng_, reinterpret_cast<const int64_t*>(nodes.data()), num_nodes_), "csl_set_nodes");
        int inflight_slot = -1, inflight_n = 0;
        for (long r = 0; r <= rounds && !stop_ && alive; r++) {
          int slot = (int)(r & 1), nb = 0;
          if (r < rounds) {
            nb = (int)std::min<long>(workers_, per_epoch - r * workers_);
            check(csl_submit_round(eng_, r * workers_, batch_, nb, slot), "csl_submit_round");
          }
          // hand out the previous round while the GPU works on this one
          for (int s = 0; s < inflight_n && !stop_ && alive; s++) {
            Staged st;
            fetch(inflight_slot, s, &st);
            alive = hand_over(st);
          }
          inflight_slot = slot;
          inflight_n = nb;
        }
      }
    } catch (const std::exception& ex) {
      {
        std::lock_guard<std::mutex> lk(cv_m_);
        conv_stop_ = true;
        cv_.notify_all();
      }
      conv.join();
      ready_.close(ex.what());
      return;
    }
    {
      std::lock_guard<std::mutex> lk(cv_m_);
      conv_done_ = true;
      cv_.notify_all();
    }
    conv.join();
  }

  // helpers of the converter thread (int32 -> long widening of a sample's 12 BiPartites); CSLICER_CONVERT_THREADS
  static int convert_helpers() {
    const char* e = getenv("CSLICER_CONVERT_THREADS");
    // default: half the CPUs, 3..8 (16 CPUs: 5.5 k samples/s with 3 helpers, 7.1-7.4 k with 8; beyond that the consumer's
    // own deletes of the previous sample set the pace)
    int def = (int)std::thread::hardware_concurrency() / 2;
    def = def < 3 ? 3 : (def > 8 ? 8 : def);
    const int n = e ? atoi(e) : def;
    return n < 0 ? 0 : (n > 15 ? 15 : n);
  }
  MiniPool pool_{convert_helpers()};
  std::mutex cv_m_;
  std::condition_variable cv_;
  Staged staged_;
  bool have_staged_ = false, conv_done_ = false, conv_stop_ = false;
};

py::list testlist(py::list l) {  // pyfrontend.cpp:94-109
  std::vector<int> v = {1, 2, 3, 4};
  py::list out = py::cast(v);
  l.append(10);  // py::list is a handle to the CALLER's list: the append is visible to the caller (pyfrontend.cpp:107)
  return out;
}

PySample* testpysample() { return empty_sample(3, 4); }  // pyfrontend.cpp:111-114

}  // namespace

PYBIND11_MODULE(cslicer, m) {
  m.doc() = "MI355X-native cslicer: drop-in for the reference pybind11 module";
  m.def("test_list", &testlist, "List testing ");
  m.def("test_pyfront", &testpysample, "List testing ", py::return_value_policy::take_ownership);
  py::class_<PySample>(m, "sample").def_readwrite("layers", &PySample::layers);
  py::class_<PyBipartite>(m, "bipatite")
      .def_readwrite("in_nodes", &PyBipartite::in_nodes)
      .def_readwrite("indptr", &PyBipartite::indptr)
      .def_readwrite("out_nodes", &PyBipartite::out_nodes)
      .def_readwrite("owned_out_nodes", &PyBipartite::owned_out_nodes)
      .def_readwrite("indices", &PyBipartite::indices)
      .def_readwrite("from_ids", &PyBipartite::from_ids)
      .def_readwrite("to_ids", &PyBipartite::to_ids)
      .def_readwrite("self_ids_in", &PyBipartite::self_ids_in)
      .def_readwrite("self_ids_out", &PyBipartite::self_ids_out)
      .def_readwrite("gpu_id", &PyBipartite::gpu_id);
  py::class_<CSlicer>(m, "cslicer")
      .def(py::init<const std::string&, int, int, int, int, const std::string&, const std::vector<int>&, int, int,
                    unsigned, bool, const std::string&>(),
           py::arg("name"), py::arg("queue_size"), py::arg("no_worker_threads"), py::arg("number_of_epochs"),
           py::arg("minibatch_size"), py::arg("data_root") = std::string(),
           py::arg("fanout") = std::vector<int>{10, 10, 10}, py::arg("n_parts") = 4, py::arg("device") = 0,
           py::arg("seed") = 5489u, py::arg("shuffle") = true, py::arg("partition") = std::string("mod"))
      .def("getSample", &CSlicer::getSample, py::return_value_policy::take_ownership,
           py::call_guard<py::gil_scoped_release>())
      .def("getNoSamples", &CSlicer::expected_number_of_samples);
}
