// hermespy_rt -- Python surface of the compute_paths hot path (pybind11).
//
// Same module name, function signature, keyword names and ChannelInfo attributes as the
// reference's binding (compute_paths_pybind11.cpp:99-210), so `import hermespy_rt` /
// `from .rt import compute_paths` users switch by putting this module on the path:
//
//   compute_paths(mesh_filepath, rx_positions, tx_positions, rx_velocities, tx_velocities,
//                 carrier_frequency, num_rx, num_tx, num_paths, num_bounces)
//       -> (los: ChannelInfo, scatter: ChannelInfo)
//   ChannelInfo.num_paths                  int   (1 / num_bounces*num_paths)
//   ChannelInfo.directions_rx/_tx          float32  (num_rx, num_tx, num_paths, 3)
//   ChannelInfo.a_te / a_tm                complex64 (num_rx, num_tx, num_paths)
//   ChannelInfo.tau / freq_shift           float32  (num_rx, num_tx, num_paths)
// The scatter path axis is bounce*num_paths + path.
//
// Deliberate differences from the reference binding (its defects):
//   * C linkage is declared on every platform (the reference only does under _WIN32 and
//     fails to import on Linux);
//   * slots the tracer does not write (dead rays, blocked records' directions, the scatter
//     directions_tx) read 0 instead of uninitialised heap memory;
//   * the RaysInfo buffers the reference allocates (too small, Q13) and then throws away are
//     not produced at all;
//   * an unreadable scene file raises ValueError instead of exit(8); tracer errors raise
//     RuntimeError; the GIL is released while the GPU works;
//   * the complex amplitudes are written in place (hrt_compute_paths_interleaved) and the arrays
//     start as untouched zero pages: the reference's binding fills four planes and interleaves them
//     afterwards (:44-97) -- on C3 those passes were 0.45 s around a 0.04 s call.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <sys/mman.h>

#include "hermespy_rt.h"

namespace py = pybind11;

namespace {

using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;

struct ChannelInfoPy {
    size_t num_paths = 0;
    py::array_t<float> directions_rx, directions_tx;
    py::array_t<std::complex<float>> a_te, a_tm;
    py::array_t<float> tau, freq_shift;
};

const Vec3 *as_vec3(const farr &a, size_t n, const char *name)
{
    if ((size_t)a.size() != 3 * n)
        throw std::invalid_argument(std::string(name) + ": expected " + std::to_string(n) +
                                    " x 3 values");
    return reinterpret_cast<const Vec3 *>(a.data());
}

// Output arrays are fresh zero pages behind a capsule -- calloc() for small ones, an anonymous mapping
// advised to use huge pages for big ones (C3: 2.3 GB): "reads 0 where the reference writes nothing"
// then costs no pass over the memory, the pages are first touched by the threads of the dense writer
// that fill them, and with 2 MiB pages those first touches are ~10^3 faults instead of ~6 10^5
// (py::array_t + memset was 0.3 s of a 0.5 s call on C3).
struct Mapping { void *p; size_t bytes; };
template <typename T>
py::array_t<T> zeros(std::vector<size_t> shape)
{
    size_t n = 1;
    for (size_t d : shape) n *= d;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    if (bytes >= ((size_t)4 << 20)) {
        const size_t huge = (size_t)2 << 20, len = (bytes + huge - 1) & ~(huge - 1);
        void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(p, len, MADV_HUGEPAGE);   // a hint; plain pages are as correct
        py::capsule owner(new Mapping{p, len}, [](void *q) {
            Mapping *m = static_cast<Mapping *>(q);
            munmap(m->p, m->bytes);
            delete m;
        });
        return py::array_t<T>(shape, static_cast<T *>(p), owner);
    }
    T *p = static_cast<T *>(std::calloc(n ? n : 1, sizeof(T)));
    if (!p) throw std::bad_alloc();
    py::capsule owner(p, [](void *q) { std::free(q); });
    return py::array_t<T>(shape, p, owner);
}

void check_scene_file(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw py::value_error("cannot open scene file: " + path);
    char magic[3] = {0, 0, 0};
    const size_t got = std::fread(magic, 1, 3, f);
    std::fclose(f);
    if (got != 3 || std::memcmp(magic, "HRT", 3) != 0)
        throw py::value_error("not an HRT scene file: " + path);
}

// one channel block: every output is written in place by the library -- the complex amplitudes too,
// through hrt_compute_paths_interleaved (re at [2 i], im at [2 i + 1] of the complex64 arrays)
struct ChannelBuffers {
    ChannelInfoPy out;
    ChannelInfo c{};
    ChannelBuffers(size_t nrx, size_t ntx, size_t n)
    {
        out.num_paths = n;
        out.directions_rx = zeros<float>({nrx, ntx, n, 3});
        out.directions_tx = zeros<float>({nrx, ntx, n, 3});
        out.a_te = zeros<std::complex<float>>({nrx, ntx, n});
        out.a_tm = zeros<std::complex<float>>({nrx, ntx, n});
        out.tau = zeros<float>({nrx, ntx, n});
        out.freq_shift = zeros<float>({nrx, ntx, n});
        c.num_rays = (uint32_t)n;
        c.directions_rx = reinterpret_cast<Vec3 *>(out.directions_rx.mutable_data());
        c.directions_tx = reinterpret_cast<Vec3 *>(out.directions_tx.mutable_data());
        float *te = reinterpret_cast<float *>(out.a_te.mutable_data());
        float *tm = reinterpret_cast<float *>(out.a_tm.mutable_data());
        c.a_te_re = te; c.a_te_im = te + 1;
        c.a_tm_re = tm; c.a_tm_im = tm + 1;
        c.tau = out.tau.mutable_data();
        c.freq_shift = out.freq_shift.mutable_data();
    }
};

std::tuple<ChannelInfoPy, ChannelInfoPy> compute_paths_py(
    const std::string &mesh_filepath, farr rx_positions, farr tx_positions, farr rx_velocities,
    farr tx_velocities, float carrier_frequency, unsigned long num_rx, unsigned long num_tx,
    unsigned long num_paths, unsigned long num_bounces)
{
    if (!num_rx || !num_tx || !num_paths || !num_bounces)
        throw std::invalid_argument("num_rx, num_tx, num_paths, num_bounces must be > 0");
    const Vec3 *rxp = as_vec3(rx_positions, num_rx, "rx_positions");
    const Vec3 *txp = as_vec3(tx_positions, num_tx, "tx_positions");
    const Vec3 *rxv = as_vec3(rx_velocities, num_rx, "rx_velocities");
    const Vec3 *txv = as_vec3(tx_velocities, num_tx, "tx_velocities");
    check_scene_file(mesh_filepath);

    ChannelBuffers los(num_rx, num_tx, 1), scat(num_rx, num_tx, num_bounces * num_paths);
    int rc;
    std::string err;
    {
        py::gil_scoped_release nogil;
        Scene scene = scene_load(mesh_filepath.c_str());
        rc = hrt_compute_paths_interleaved(&scene, rxp, txp, rxv, txv, carrier_frequency, num_rx, num_tx,
                                           num_paths, num_bounces, &los.c, nullptr, &scat.c, nullptr,
                                           nullptr);
        if (rc != HRT_OK) err = hrt_last_error();
        free_scene(&scene);
    }
    if (rc != HRT_OK)
        throw std::runtime_error("hermespy_rt.compute_paths failed (" + std::to_string(rc) +
                                 "): " + err);
    return {std::move(los.out), std::move(scat.out)};
}

// compute_paths_list: the same call with the result as ONE list of path records (extension; the
// reference has no such form).  Returns a dict of numpy arrays, entry n being the record the dense
// form holds at [rx[n], tx[n], bounce[n] * num_paths + path[n]].
template <typename T>
py::array_t<T> copy_out(const T *src, std::vector<size_t> shape)
{
    py::array_t<T> a(shape);
    if (a.size()) std::memcpy(a.mutable_data(), src, sizeof(T) * (size_t)a.size());
    return a;
}

py::dict compute_paths_list_py(const std::string &mesh_filepath, farr rx_positions, farr tx_positions,
                               farr rx_velocities, farr tx_velocities, float carrier_frequency,
                               unsigned long num_rx, unsigned long num_tx, unsigned long num_paths,
                               unsigned long num_bounces, bool include_blocked)
{
    if (!num_rx || !num_tx || !num_paths || !num_bounces)
        throw std::invalid_argument("num_rx, num_tx, num_paths, num_bounces must be > 0");
    const Vec3 *rxp = as_vec3(rx_positions, num_rx, "rx_positions");
    const Vec3 *txp = as_vec3(tx_positions, num_tx, "tx_positions");
    const Vec3 *rxv = as_vec3(rx_velocities, num_rx, "rx_velocities");
    const Vec3 *txv = as_vec3(tx_velocities, num_tx, "tx_velocities");
    check_scene_file(mesh_filepath);
    // the list stays where the library built it: every numpy array is a view of one of its fields
    // and keeps it alive through one shared capsule (copying 1.3 GB of fields was 0.19 s of a
    // 0.24 s call on C3); only the complex amplitudes are formed here, from the re/im planes
    hrt_path_list *plp = new hrt_path_list;
    std::memset(plp, 0, sizeof *plp);
    py::capsule owner(plp, [](void *q) {
        hrt_path_list *p = static_cast<hrt_path_list *>(q);
        hrt_path_list_free(p);
        delete p;
    });
    hrt_path_list &pl = *plp;
    int rc;
    std::string err;
    {
        py::gil_scoped_release nogil;
        Scene scene = scene_load(mesh_filepath.c_str());
        rc = hrt_compute_paths_list(&scene, rxp, txp, rxv, txv, carrier_frequency, num_rx, num_tx,
                                    num_paths, num_bounces, include_blocked ? 1 : 0, &pl, nullptr);
        if (rc != HRT_OK) err = hrt_last_error();
        free_scene(&scene);
    }
    if (rc != HRT_OK)
        throw std::runtime_error("hermespy_rt.compute_paths_list failed (" + std::to_string(rc) +
                                 "): " + err);
    const size_t n = (size_t)pl.num;
    py::array_t<std::complex<float>> te = zeros<std::complex<float>>({n}), tm = zeros<std::complex<float>>({n});
    {
        py::gil_scoped_release nogil;
        std::complex<float> *pte = te.mutable_data(), *ptm = tm.mutable_data();
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(8, n / 1000000));
        std::vector<std::thread> th;
        auto part = [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; ++i) {
                pte[i] = {pl.a_te_re[i], pl.a_te_im[i]};
                ptm[i] = {pl.a_tm_re[i], pl.a_tm_im[i]};
            }
        };
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(part, n * k / nt, n * (k + 1) / nt);
        part(0, n / nt);
        for (auto &x : th) x.join();
    }
    auto view = [&](auto *ptr, std::vector<size_t> shape) {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(ptr)>>;
        return py::array_t<T>(shape, ptr, owner);
    };
    py::dict d;
    d["rx"] = view(pl.rx, {n});
    d["tx"] = view(pl.tx, {n});
    d["bounce"] = view(pl.bounce, {n});
    d["path"] = view(pl.path, {n});
    d["a_te"] = te;
    d["a_tm"] = tm;
    d["tau"] = view(pl.tau, {n});
    d["direction_rx"] = view(reinterpret_cast<float *>(pl.direction_rx), {n, 3});
    d["freq_shift"] = view(pl.freq_shift, {n});
    d["unblocked"] = view(reinterpret_cast<bool *>(pl.unblocked), {n});
    d["mesh"] = view(pl.mesh, {n});
    d["face"] = view(pl.face, {n});
    d["los"] = view(pl.los, {(size_t)num_rx, (size_t)num_tx, (size_t)8});
    return d;
}

}  // namespace

PYBIND11_MODULE(hermespy_rt, m)
{
    m.doc() = "MI355X-native compute_paths (drop-in for the hermespy-rt binding)";
    py::class_<ChannelInfoPy>(m, "ChannelInfo")
        .def_readonly("num_paths", &ChannelInfoPy::num_paths)
        .def_readonly("directions_rx", &ChannelInfoPy::directions_rx)
        .def_readonly("directions_tx", &ChannelInfoPy::directions_tx)
        .def_readonly("a_te", &ChannelInfoPy::a_te)
        .def_readonly("a_tm", &ChannelInfoPy::a_tm)
        .def_readonly("tau", &ChannelInfoPy::tau)
        .def_readonly("freq_shift", &ChannelInfoPy::freq_shift);
    m.def("compute_paths", &compute_paths_py, "Compute gains and delays",
          py::arg("mesh_filepath"), py::arg("rx_positions"), py::arg("tx_positions"),
          py::arg("rx_velocities"), py::arg("tx_velocities"), py::arg("carrier_frequency"),
          py::arg("num_rx"), py::arg("num_tx"), py::arg("num_paths"), py::arg("num_bounces"));
    m.def("compute_paths_list", &compute_paths_list_py,
          "compute_paths with the result as one list of path records (dict of arrays)",
          py::arg("mesh_filepath"), py::arg("rx_positions"), py::arg("tx_positions"),
          py::arg("rx_velocities"), py::arg("tx_velocities"), py::arg("carrier_frequency"),
          py::arg("num_rx"), py::arg("num_tx"), py::arg("num_paths"), py::arg("num_bounces"),
          py::arg("include_blocked") = false);
    m.def("version", []() { return std::string(hrt_version()); });
    // Between calls the library keeps the device workspace and the page-locked staging of the last
    // call (C3: 3.3 GB of HBM, 0.4 GB of pinned host memory; up to HRT_POOL_MAX_BYTES, default 24 GiB)
    // and parked helper threads: this gives them back (csrc/host/compute_paths.c, hrt_cache_clear).
    m.def("cache_clear", []() { hrt_cache_clear(); },
          "Release the device / pinned buffers and helper threads kept between compute_paths calls");
}
