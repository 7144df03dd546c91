// window_capi.cpp — C ABI (include/visfs_window.h) over VISFS::WindowMap.
#include <cstring>
#include <exception>
#include <map>
#include <set>
#include <string>

#include "../../include/visfs_window.h"
#include "WindowMap.h"

struct visfs_window_map { VISFS::WindowMap map; explicit visfs_window_map(const std::map<std::string, std::string>& p) : map(p) {} };

extern "C" {

int visfs_window_abi_version(void) { return VISFS_WINDOW_ABI_VERSION; }

int visfs_window_create(int n, const char* const* keys, const char* const* values, visfs_window_map** out) {
    if (!out || n < 0 || (n > 0 && (!keys || !values))) return VISFS_BA_ERR_BAD_ARGUMENT;
    try {
        std::map<std::string, std::string> p;
        for (int i = 0; i < n; ++i) p[keys[i]] = values[i];
        *out = new visfs_window_map(p);
        return VISFS_BA_OK;
    } catch (const std::exception&) { return VISFS_BA_ERR_BAD_ARGUMENT; }
}

void visfs_window_destroy(visfs_window_map* m) { delete m; }

int visfs_window_insert(visfs_window_map* m, uint64_t id, const double pose[12], const double wheel[12], const double translation[3],
                        int32_t nw, const uint64_t* wid, const float* wuv, const float* wxyz, const uint8_t* has3d,
                        int32_t nc, const uint64_t* cid, const float* cuv) {
    if (!m || !pose || !wheel || !translation || nw < 0 || nc < 0 || (nw && (!wid || !wuv || !wxyz || !has3d)) || (nc && (!cid || !cuv))) return -VISFS_BA_ERR_BAD_ARGUMENT;
    for (int i = 1; i < nw; ++i) if (wid[i] <= wid[i - 1]) return -VISFS_BA_ERR_BAD_ARGUMENT;
    for (int i = 1; i < nc; ++i) if (cid[i] <= cid[i - 1]) return -VISFS_BA_ERR_BAD_ARGUMENT;
    try {
        VISFS::SignatureInput s;
        s.id = id;
        std::memcpy(s.pose, pose, 96); std::memcpy(s.wheelOdom, wheel, 96);
        s.words.resize(nw);
        for (int i = 0; i < nw; ++i) s.words[i] = { wid[i], wuv[4 * i], wuv[4 * i + 1], wuv[4 * i + 2], wuv[4 * i + 3], wxyz[3 * i], wxyz[3 * i + 1], wxyz[3 * i + 2], has3d[i] != 0 };
        s.covisibleWords.resize(nc);
        for (int i = 0; i < nc; ++i) s.covisibleWords[i] = { cid[i], cuv[2 * i], cuv[2 * i + 1] };
        return m->map.insertSignature(s, translation) ? 1 : 0;
    } catch (const std::exception&) { return -VISFS_BA_ERR_BAD_ARGUMENT; }
}

void visfs_window_remove(visfs_window_map* m) { if (m) m->map.removeSignature(); }
int visfs_window_available(const visfs_window_map* m) { return m && m->map.checkMapAvaliable() ? 1 : 0; }
int visfs_window_is_key_signature(const visfs_window_map* m) { return m && m->map.isKeySignature() ? 1 : 0; }

int visfs_window_build(visfs_window_map* m, const double Trc[12], double fx, double fy, double cx, double cy, float baseline,
                       int32_t nCameras, int32_t withLinks, visfs_ba_window* out) {
    if (!m || !Trc || !out) return VISFS_BA_ERR_BAD_ARGUMENT;
    try { *out = m->map.buildWindow(Trc, fx, fy, cx, cy, baseline, nCameras, withLinks != 0); return VISFS_BA_OK; }
    catch (const std::exception&) { return VISFS_BA_ERR_BAD_ARGUMENT; }
}

static int emitErrors(const std::set<uint64_t>& ev, uint64_t* out, int32_t capacity, int32_t* nError) {
    int32_t k = 0;
    for (uint64_t id : ev) { if (k < capacity && out) out[k] = id; ++k; }
    if (nError) *nError = k;
    return VISFS_BA_OK;
}

int visfs_window_update(visfs_window_map* m, int32_t np, const uint64_t* pid, const double* pT, int32_t npt, const uint64_t* ptid, const double* pxyz,
                        int32_t no, const uint64_t* of, const uint64_t* op, uint64_t* errorVertex, int32_t capacity, int32_t* nError) {
    if (!m || np < 0 || npt < 0 || no < 0 || (np && (!pid || !pT)) || (npt && (!ptid || !pxyz)) || (no && (!of || !op))) return VISFS_BA_ERR_BAD_ARGUMENT;
    try {
        std::set<uint64_t> ev;
        m->map.updateLocalMap(np, pid, pT, npt, ptid, pxyz, no, of, op, ev);
        return emitErrors(ev, errorVertex, capacity, nError);
    } catch (const std::exception&) { return VISFS_BA_ERR_BAD_ARGUMENT; }
}

int visfs_window_apply(visfs_window_map* m, const visfs_ba_result* r, uint64_t* errorVertex, int32_t capacity, int32_t* nError) {
    if (!m || !r) return VISFS_BA_ERR_BAD_ARGUMENT;
    try {
        std::set<uint64_t> ev;
        m->map.applyResult(*r, ev);
        return emitErrors(ev, errorVertex, capacity, nError);
    } catch (const std::exception&) { return VISFS_BA_ERR_BAD_ARGUMENT; }
}

int visfs_window_counts(const visfs_window_map* m, int32_t* ns, int32_t* nf, int32_t* no) {
    if (!m) return VISFS_BA_ERR_BAD_ARGUMENT;
    const auto f = m->map.features();
    int32_t obs = 0;
    for (const auto& v : f) obs += static_cast<int32_t>(v.obsSignature.size());
    if (ns) *ns = static_cast<int32_t>(m->map.signatureCount());
    if (nf) *nf = static_cast<int32_t>(f.size());
    if (no) *no = obs;
    return VISFS_BA_OK;
}

int visfs_window_counters(const visfs_window_map* m, int32_t* nf, int32_t* ns, float* parallax, double translation[3]) {
    if (!m || !nf || !ns || !parallax || !translation) return VISFS_BA_ERR_BAD_ARGUMENT;
    int a, b;
    m->map.counters(a, b, *parallax, translation);
    *nf = a; *ns = b;
    return VISFS_BA_OK;
}

int visfs_window_dump(const visfs_window_map* m, uint64_t* sigIds, double* sigPose, uint64_t* fid, uint64_t* fstart, uint64_t* fend,
                      int32_t* fstate, double* fxyz, int32_t* fnobs, uint64_t* obsSig, float* obsVals) {
    if (!m) return VISFS_BA_ERR_BAD_ARGUMENT;
    const auto ids = m->map.signatureIds();
    for (std::size_t i = 0; i < ids.size(); ++i) { if (sigIds) sigIds[i] = ids[i]; if (sigPose) m->map.signaturePose(i, sigPose + 12 * i); }
    const auto f = m->map.features();
    std::size_t o = 0;
    for (std::size_t i = 0; i < f.size(); ++i) {
        if (fid) fid[i] = f[i].id;
        if (fstart) fstart[i] = f[i].startSignature;
        if (fend) fend[i] = f[i].endSignature;
        if (fstate) fstate[i] = f[i].state;
        if (fxyz) std::memcpy(fxyz + 3 * i, f[i].pose, 24);
        if (fnobs) fnobs[i] = static_cast<int32_t>(f[i].obsSignature.size());
        for (std::size_t k = 0; k < f[i].obsSignature.size(); ++k, ++o) {
            if (obsSig) obsSig[o] = f[i].obsSignature[k];
            if (obsVals) std::memcpy(obsVals + 7 * o, &f[i].obs[7 * k], 28);
        }
    }
    return VISFS_BA_OK;
}

}  // extern "C"
