// Sparsity pattern builder (host, multi-threaded).  No reference counterpart: the reference
// hands COO triplets to scipy (fcVM.py:1111); here the element -> matrix-slot map is fixed once
// per mesh so that assembly is a pure scatter and SpMV streams a wavefront-shaped layout.
#include "fcvm_pattern.h"

#include <algorithm>
#include <functional>
#include <stdexcept>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace fcvm {

void parallel_for(int64_t n, int nthreads, const std::function<void(int64_t, int64_t, int)> &fn, int64_t min_parallel) {
    if (nthreads < 1) nthreads = 1;
    if (n < min_parallel || nthreads == 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    int64_t chunk = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int64_t b = t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        th.emplace_back(fn, b, e, t);
    }
    for (auto &x : th) x.join();
}

void build_pattern(int64_t ne, int64_t nn, const int32_t *el, Pattern &p, int nthreads, int sort_window, int npe,
                   const int32_t *row_order, bool want_emap) {
    const bool timing = getenv("FCVM_PATTERN_TIMING") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "fcvm pattern: %-28s %.3f s\n", what, std::chrono::duration<double>(t - tprev).count());
        tprev = t;
    };
    if (ne <= 0 || nn <= 0) throw std::runtime_error("empty mesh");
    for (int64_t i = 0; i < ne * npe; i++)
        if (el[i] < 0 || el[i] >= nn) throw std::runtime_error("element node index out of range");
    p.ne = ne;
    p.nn = nn;
    p.npe = npe;
    // node -> element incidence
    if (ne >= (int64_t(1) << 27)) throw std::runtime_error("more than 2^27 elements per device are not supported");
    std::vector<int64_t> &nptr = p.nptr;
    RawVec<int32_t> &nel = p.nel;   // element << 4 | local node index, ascending element order per node
    int nt = std::max(1, nthreads);
    nptr.assign(nn + 1, 0);
    nel.resize(ne * npe);
    if (ne * npe < (int64_t)1 << 20 || nt == 1) {
        for (int64_t i = 0; i < ne * npe; i++) nptr[el[i] + 1]++;
        for (int64_t i = 0; i < nn; i++) nptr[i + 1] += nptr[i];
        std::vector<int64_t> fill(nptr.begin(), nptr.end() - 1);
        for (int64_t e = 0; e < ne; e++)
            for (int k = 0; k < npe; k++) nel[fill[el[e * npe + k]]++] = (int32_t)((e << 4) | k);
    } else {
        // large meshes: counts and placement with atomic cursors on all threads, then every node's short list is sorted --
        // the same ascending element order as the serial loop gives, whatever the threads' interleaving was
        std::vector<std::atomic<int32_t>> cnt(nn);
        parallel_for(nn, nt, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) cnt[i].store(0, std::memory_order_relaxed); });
        parallel_for(ne, nt, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b * npe; i < e * npe; i++) cnt[el[i]].fetch_add(1, std::memory_order_relaxed);
        });
        for (int64_t i = 0; i < nn; i++) nptr[i + 1] = nptr[i] + cnt[i].load(std::memory_order_relaxed);
        parallel_for(nn, nt, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) cnt[i].store(0, std::memory_order_relaxed); });
        parallel_for(ne, nt, [&](int64_t b, int64_t e, int) {
            for (int64_t k = b; k < e; k++)
                for (int a = 0; a < npe; a++) {
                    const int32_t nd = el[k * npe + a];
                    nel[nptr[nd] + cnt[nd].fetch_add(1, std::memory_order_relaxed)] = (int32_t)((k << 4) | a);
                }
        });
        parallel_for(nn, nt, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; i++) std::sort(nel.begin() + nptr[i], nel.begin() + nptr[i + 1]);
        });
    }
    lap("node -> elements");
    // node -> sorted unique neighbour nodes (diagonal included)
    std::vector<std::vector<int32_t>> chunks(nt);
    std::vector<int64_t> len(nn, 0);
    std::vector<int64_t> cbeg(nt, 0), cend(nt, 0);
    parallel_for(nn, nt, [&](int64_t b, int64_t e, int t) {
        cbeg[t] = b;
        cend[t] = e;
        std::vector<int32_t> tmp;
        std::vector<int32_t> stamp(nn, -1);   // row that last listed a node: the ~140 candidates of a row shrink to its ~28
                                              // distinct neighbours before anything is sorted
        auto &out = chunks[t];
        for (int64_t i = b; i < e; i++) {
            tmp.clear();
            for (int64_t q = nptr[i]; q < nptr[i + 1]; q++) {
                const int32_t *nd = el + (int64_t)(nel[q] >> 4) * npe;
                for (int k = 0; k < npe; k++)
                    if (stamp[nd[k]] != (int32_t)i) { stamp[nd[k]] = (int32_t)i; tmp.push_back(nd[k]); }
            }
            if (tmp.empty()) tmp.push_back((int32_t)i);  // isolated node: keep a diagonal block
            std::sort(tmp.begin(), tmp.end());
            len[i] = (int64_t)tmp.size();
            out.insert(out.end(), tmp.begin(), tmp.end());
        }
    });
    lap("node adjacency");
    p.rowptr.assign(nn + 1, 0);
    for (int64_t i = 0; i < nn; i++) p.rowptr[i + 1] = p.rowptr[i] + len[i];
    p.nnzb = p.rowptr[nn];
    p.colidx.resize(p.nnzb);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            if (cend[t] > cbeg[t])
                th.emplace_back([&, t]() { std::copy(chunks[t].begin(), chunks[t].end(), p.colidx.begin() + p.rowptr[cbeg[t]]); });
        for (auto &x : th) x.join();
    }
    chunks.clear();
    chunks.shrink_to_fit();
    lap("CSR copy");
    // slices: rows sorted by descending length inside windows of SORT_WINDOW slices
    p.nslices = (nn + SLICE - 1) / SLICE;
    p.rownode.assign(p.nslices * SLICE, -1);
    p.slotof.assign(nn, -1);
    const int64_t win = (int64_t)SLICE * sort_window;
    {
        const int64_t nwin = (nn + win - 1) / win;
        auto sort_windows = [&](int64_t b, int64_t e, int) {
            std::vector<int32_t> ids;
            for (int64_t wi = b; wi < e; wi++) {
                const int64_t w0 = wi * win, w1 = std::min(nn, w0 + win);
                ids.resize(w1 - w0);
                for (int64_t i = w0; i < w1; i++) ids[i - w0] = row_order ? row_order[i] : (int32_t)i;
                std::stable_sort(ids.begin(), ids.end(), [&](int32_t a, int32_t b2) { return len[a] > len[b2]; });
                for (int64_t k = 0; k < (int64_t)ids.size(); k++) {
                    p.rownode[w0 + k] = ids[k];
                    p.slotof[ids[k]] = (int32_t)(w0 + k);
                }
            }
        };
        if (nwin >= 64) {     // (parallel_for runs serially below 4096 items: split the windows by hand)
            std::vector<std::thread> th;
            const int64_t chunk = (nwin + nt - 1) / nt;
            for (int t = 0; t < nt; t++) {
                const int64_t b = t * chunk, e = std::min(nwin, b + chunk);
                if (b < e) th.emplace_back(sort_windows, b, e, t);
            }
            for (auto &x : th) x.join();
        } else {
            sort_windows(0, nwin, 0);
        }
    }
    lap("length sort");
    p.sptr.assign(p.nslices + 1, 0);
    p.maxlen = 0;
    for (int64_t s = 0; s < p.nslices; s++) {
        int64_t m = 0;
        for (int l = 0; l < SLICE; l++) {
            int32_t nd = p.rownode[s * SLICE + l];
            if (nd >= 0) m = std::max(m, len[nd]);
        }
        p.maxlen = std::max(p.maxlen, m);
        p.sptr[s + 1] = p.sptr[s] + std::max<int64_t>(m, 1);   // pairs of blocks + a single last block when m is odd
    }
    p.nslots = p.sptr[p.nslices] * SLICE;
    if ((uint64_t)p.sptr[p.nslices] >= (1ull << 30))
        throw std::runtime_error("matrix too large for the packed 32-bit block locations (2^30 half chunks)");
    lap("slice lengths");
    p.cols.resize(p.nslots);      // every entry is written below
    parallel_for(p.nslices, nt, [&](int64_t b, int64_t e, int) {
        for (int64_t s = b; s < e; s++) {
            int64_t L = p.sptr[s + 1] - p.sptr[s];
            for (int l = 0; l < SLICE; l++) {
                int32_t nd = p.rownode[s * SLICE + l];
                for (int64_t j = 0; j < L; j++) {
                    int32_t c = 0;
                    if (nd >= 0) c = (j < len[nd]) ? p.colidx[p.rowptr[nd] + j] : nd;
                    p.cols[p.col_index(s, l, j)] = c;
                }
            }
        }
    });
    lap("columns");
    // element block (a,b) -> offset of its component 0
    // walked by ROW node: the row's column list stays in the cache for the searches of all its elements (by element, every
    // one of the ten rows of an element was a cache miss: 0.096 -> see FCVM_PATTERN_TIMING)
    if (want_emap) p.emap.resize(ne * npe * npe);
    else p.emap.clear();     // the caller fills the map on the device (k_build_emap)
    if (want_emap) parallel_for(nn, nt, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            const int32_t *rb = p.colidx.data() + p.rowptr[i];
            const int32_t *re = p.colidx.data() + p.rowptr[i + 1];
            const int64_t slot = p.slotof[i];
            for (int64_t q = nptr[i]; q < nptr[i + 1]; q++) {
                const int64_t k = nel[q] >> 4;
                const int a = nel[q] & 15;
                const int32_t *nd = el + k * npe;
                uint32_t *out = p.emap.data() + (k * npe + a) * npe;
                for (int c = 0; c < npe; c++) out[c] = p.pack_offset(slot, std::lower_bound(rb, re, nd[c]) - rb);
            }
        }
    });
    lap("element map");
    p.diagoff.resize(nn);
    parallel_for(nn, nt, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            const int32_t *rb = p.colidx.data() + p.rowptr[i];
            const int32_t *re = p.colidx.data() + p.rowptr[i + 1];
            int64_t j = std::lower_bound(rb, re, (int32_t)i) - rb;
            p.diagoff[i] = p.abs_offset(p.slotof[i], j);
        }
    });
}

std::vector<int32_t> morton_order(int64_t nn, const double *xyz) {
    int nt = (int)std::thread::hardware_concurrency();
    if (const char *v = getenv("FCVM_MORTON_THREADS")) nt = atoi(v);
    nt = std::max(1, std::min(nt, 16));
    if (nn < 262144) nt = 1;
    auto on_chunks = [&](const std::function<void(int64_t, int64_t, int)> &fn) { parallel_for(nn, nt, fn, 0); };
    std::vector<double> tlo(3 * nt, 1e300), thi(3 * nt, -1e300);
    on_chunks([&](int64_t b, int64_t e, int t) {
        double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
        for (int64_t i = b; i < e; i++)
            for (int k = 0; k < 3; k++) {
                l[k] = std::min(l[k], xyz[3 * i + k]);
                h[k] = std::max(h[k], xyz[3 * i + k]);
            }
        for (int k = 0; k < 3; k++) { tlo[3 * t + k] = l[k]; thi[3 * t + k] = h[k]; }
    });
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int t = 0; t < nt; t++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], tlo[3 * t + k]);
            hi[k] = std::max(hi[k], thi[3 * t + k]);
        }
    auto spread = [](uint64_t v) {  // 21 bits -> every third bit
        v &= 0x1fffff;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    RawVec<uint64_t> key(nn), key2(nn);
    RawVec<int32_t> order(nn), order2(nn);
    // ONE scale for the three axes (the largest extent): cells of the curve are cubes in space.  Until round 3 every axis was
    // scaled by its own extent, which on a plate of 200 x 100 x 10 cuts the thickness as finely as the length: groups of
    // consecutive nodes came out as thin slabs with twice the surface of a compact patch (1 M plate: 35 % of the blocks inside a
    // group of 512 rows instead of 67 %, windows of 2000 nodes instead of ...: see DESIGN.md "SpMV").  FCVM_MORTON_AXIS_SCALE=1: the old keys.
    const bool per_axis = getenv("FCVM_MORTON_AXIS_SCALE") && atoi(getenv("FCVM_MORTON_AXIS_SCALE")) != 0;
    const double emax = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    on_chunks([&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            uint64_t q[3];
            for (int k = 0; k < 3; k++) {
                double ext = per_axis ? hi[k] - lo[k] : emax;
                q[k] = ext > 0 ? (uint64_t)((xyz[3 * i + k] - lo[k]) / ext * 2097151.0) : 0;
            }
            key[i] = spread(q[0]) | spread(q[1]) << 1 | spread(q[2]) << 2;
            order[i] = (int32_t)i;
        }
    });
    // stable LSD radix sort of the 63-bit keys, six passes of 11 bits, every pass on all threads: per-thread digit counts over
    // the thread's chunk, offsets by (digit, thread), scatter.  Equal keys keep ascending node numbers: the order
    // std::sort gives the (key, node) pairs, whatever the thread count (round 3: 0.041 -> see FCVM_PATTERN_TIMING at 1.4 M nodes)
    constexpr int BITS = 11, NB = 1 << BITS;
    std::vector<int64_t> cnt((size_t)NB * nt);
    for (int pass = 0; pass * BITS < 63; pass++) {
        const int sh = BITS * pass;
        std::fill(cnt.begin(), cnt.end(), 0);
        on_chunks([&](int64_t b, int64_t e, int t) {
            int64_t *c = cnt.data() + (size_t)t * NB;
            for (int64_t i = b; i < e; i++) c[(key[i] >> sh) & (NB - 1)]++;
        });
        int64_t at = 0;
        for (int d = 0; d < NB; d++)
            for (int t = 0; t < nt; t++) {
                const int64_t n = cnt[(size_t)t * NB + d];
                cnt[(size_t)t * NB + d] = at;
                at += n;
            }
        on_chunks([&](int64_t b, int64_t e, int t) {
            int64_t *c = cnt.data() + (size_t)t * NB;
            for (int64_t i = b; i < e; i++) {
                const int64_t to = c[(key[i] >> sh) & (NB - 1)]++;
                key2[to] = key[i];
                order2[to] = order[i];
            }
        });
        key.swap(key2);
        order.swap(order2);
    }
    return std::vector<int32_t>(order.begin(), order.end());
}

double numbering_spread(int64_t ne, int64_t nn, const int32_t *el, int npe) {
    double s = 0.0;
    for (int64_t e = 0; e < ne; e++) {
        int32_t mn = el[e * npe], mx = mn;
        for (int k = 1; k < npe; k++) {
            mn = std::min(mn, el[e * npe + k]);
            mx = std::max(mx, el[e * npe + k]);
        }
        s += (double)(mx - mn);
    }
    return ne > 0 && nn > 0 ? s / (double)ne / (double)nn : 0.0;
}

bool build_window_format(const Pattern &p, const double *xyz, int nthreads, int64_t max_window, WinPattern &w, int waves, int split,
                         const int32_t *given_order) {
    const int64_t nn = p.nn, WV = waves, R = WV * SLICE;
    w.waves = waves;
    w.split = split;
    const bool deal = getenv("FCVM_WIN_DEAL") && atoi(getenv("FCVM_WIN_DEAL")) == 1;
    if (p.maxlen > 65535) return false;
    const bool timing = getenv("FCVM_PATTERN_TIMING") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "fcvm window format: %-25s %.3f s\n", what, std::chrono::duration<double>(t - tprev).count());
        tprev = t;
    };
    std::vector<int32_t> order = given_order ? std::vector<int32_t>(given_order, given_order + nn) : morton_order(nn, xyz);
    lap("Morton order");
    auto nseg_of = [&](int32_t nd) { return (int)((p.rowptr[nd + 1] - p.rowptr[nd] + split - 1) / split); };
    // the segment count of a row is packed into 8 bits of `seg`, and a row's segments must all lie in ONE group: refuse
    // (the caller falls back to the sliced-ELL kernel) instead of corrupting the tables
    if ((p.maxlen + split - 1) / split > std::min<int64_t>(255, R)) return false;
    // groups: consecutive nodes of the Morton order while their vrows fit
    std::vector<int64_t> gfirst{0};
    {
        int64_t used = 0;
        for (int64_t i = 0; i < nn; i++) {
            const int k = std::max(1, nseg_of(order[i]));
            if (used + k > R) { gfirst.push_back(i); used = 0; }
            used += k;
        }
        gfirst.push_back(nn);
    }
    const int64_t ng = (int64_t)gfirst.size() - 1;
    w.ngroups = ng;
    w.nwaves = ng * WV;
    w.rownode.assign(w.nwaves * SLICE, -1);
    w.srcslot.assign(w.nwaves * SLICE, 0);
    w.rowlen.assign(w.nwaves * SLICE, 0);
    w.jstart.assign(w.nwaves * SLICE, 0);
    w.seg.assign(w.nwaves * SLICE, 1);
    w.own.assign(w.nwaves * SLICE, 0);
    w.woff.assign(w.nwaves + 1, 0);
    w.gptr.assign(ng + 1, 0);
    w.gextra.assign(ng, 0);
    struct VRow { int64_t len; int32_t nd; int32_t j0; uint32_t seg; };
    // pass 1: vrows of every group sorted by length, dealt round-robin to the waves; window sizes; entries per wave
    std::vector<std::vector<int32_t>> win(ng);
    parallel_for(ng, nthreads, [&](int64_t b, int64_t e, int) {
        std::vector<VRow> rows;
        std::vector<int32_t> stamp(nn, -1);     // group that last listed the node: unique window entries without a big sort
        for (int64_t g = b; g < e; g++) {
            rows.clear();
            auto &wl = win[g];
            int32_t extra = 0;
            for (int64_t i = gfirst[g]; i < gfirst[g + 1]; i++) {
                const int32_t nd = order[i];
                const int64_t len = p.rowptr[nd + 1] - p.rowptr[nd];
                const int k = std::max(1, nseg_of(nd));
                int64_t j0 = 0;
                for (int t = 0; t < k; t++) {       // equal segments: len = q * k + rem, the first rem segments one longer
                    const int64_t l = len / k + (t < len % k ? 1 : 0);
                    const uint32_t code = t == 0 ? ((uint32_t)extra << 8) | (uint32_t)k : ((uint32_t)(extra + t - 1) << 8);
                    rows.push_back({l, nd, (int32_t)j0, code});
                    j0 += l;
                }
                extra += k - 1;
                for (int64_t q = p.rowptr[nd]; q < p.rowptr[nd + 1]; q++) {
                    const int32_t col = p.colidx[q];
                    if (stamp[col] != (int32_t)g) { stamp[col] = (int32_t)g; wl.push_back(col); }
                }
            }
            w.gextra[g] = extra;
            std::stable_sort(rows.begin(), rows.end(), [](const VRow &x, const VRow &y) { return x.len > y.len; });
            for (size_t k = 0; k < rows.size(); k++) {
                // consecutive chunks of the sorted list: uniform lengths inside a wave, few partly filled block rows (0.482 against
                // 0.490 ms for round-robin dealing, which balances the waves of a group but makes every block row ragged:
                // FCVM_WIN_DEAL=1, profiles/r02ze_spmv_window_dealing_ab.txt)
                const int64_t wave = deal ? g * WV + (int64_t)k % WV : g * WV + (int64_t)k / SLICE;
                const int64_t lane = deal ? (int64_t)k / WV : (int64_t)k % SLICE;
                const int64_t at = wave * SLICE + lane;
                w.rownode[at] = rows[k].nd;
                w.srcslot[at] = p.slotof[rows[k].nd];
                w.rowlen[at] = (uint16_t)rows[k].len;
                w.jstart[at] = (uint16_t)rows[k].j0;
                w.seg[at] = rows[k].seg;
                w.woff[wave + 1] += rows[k].len;
            }
            std::sort(wl.begin(), wl.end());
            wl.shrink_to_fit();
            w.gptr[g + 1] = (int64_t)wl.size();
        }
    }, 16);
    lap("pass 1 (rows, windows)");
    w.wmax = 0;
    w.xmax = 0;
    for (int64_t g = 0; g < ng; g++) {
        w.wmax = std::max(w.wmax, w.gptr[g + 1]);
        w.xmax = std::max<int64_t>(w.xmax, w.gextra[g]);
        w.gptr[g + 1] += w.gptr[g];
    }
    w.wmean = (double)w.gptr[ng] / (double)ng;
    if (w.wmax > max_window || w.wmax > 65535 || w.xmax >= (1 << 24)) return false;
    // every wave starts on a 128-byte line of the value array (9 * 16 entries * 8 B = 9 lines): its full block rows
    // (64 lanes: 36 lines) stay aligned; costs < 16 entries per wave
    for (int64_t i = 0; i < w.nwaves; i++) w.woff[i + 1] = (w.woff[i + 1] + 15) / 16 * 16 + w.woff[i];
    w.nblocks = w.woff[w.nwaves];
    w.wlist.resize(w.gptr[ng]);
    w.cols.assign(w.nblocks, 0);
    // pass 2: window lists, 16-bit columns in the prefix layout
    parallel_for(ng, nthreads, [&](int64_t b, int64_t e, int) {
        std::vector<uint16_t> pos(nn, 0);       // window position of a node (valid for the nodes of the current window only)
        for (int64_t g = b; g < e; g++) {
            const auto &wl = win[g];
            std::copy(wl.begin(), wl.end(), w.wlist.begin() + w.gptr[g]);
            for (size_t i = 0; i < wl.size(); i++) pos[wl[i]] = (uint16_t)i;
            for (int64_t wave = g * WV; wave < (g + 1) * WV; wave++) {
                const uint16_t *len = &w.rowlen[wave * SLICE];
                int64_t off = w.woff[wave];
                for (int64_t j = 0; j < len[0]; j++) {     // lane 0 holds the longest vrow of the wave
                    int64_t n = 0;
                    while (n < SLICE && len[n] > j) n++;   // lanes sorted by length: a prefix
                    for (int64_t l = 0; l < n; l++) {
                        const int32_t nd = w.rownode[wave * SLICE + l];
                        const int32_t col = p.colidx[p.rowptr[nd] + w.jstart[wave * SLICE + l] + j];
                        w.cols[off + l] = pos[col];
                    }
                    off += n;
                }
                for (int64_t l = 0; l < SLICE; l++) {
                    const int32_t nd = w.rownode[wave * SLICE + l];
                    if (nd >= 0) w.own[wave * SLICE + l] = pos[nd];
                }
            }
        }
    }, 16);
    lap("pass 2 (columns)");
    return true;
}

// group of every node = the group whose first-segment vrow carries it
static std::vector<int32_t> window_group_of(const Pattern &p, const WinPattern &w) {
    std::vector<int32_t> grp(p.nn, -1);
    for (int64_t at = 0; at < w.nwaves * SLICE; at++)
        if (w.rownode[at] >= 0 && (w.seg[at] & 255u) > 0) grp[w.rownode[at]] = (int32_t)(at / (SLICE * w.waves));
    return grp;
}
// is block (i, j), i != j, both in one group, stored in row i?
static inline bool sym_keeps(int32_t i, int32_t j) {
    const int32_t a = std::min(i, j), b = std::max(i, j);
    return (((a + b) & 1) == 0 ? a : b) == i;
}

bool build_window_sym(const Pattern &p, const WinPattern &w, int nthreads, WinSym &s) {
    const int64_t nn = p.nn, WV = w.waves, R = WV * SLICE, ng = w.ngroups;
    const int split = w.split;
    if (p.maxlen > 255) return false;
    s.ngroups = ng; s.nwaves = w.nwaves; s.waves = w.waves; s.split = split;
    s.rownode.assign(s.nwaves * SLICE, -1);
    s.srcslot.assign(s.nwaves * SLICE, 0);
    s.rowlen.assign(s.nwaves * SLICE, 0);
    s.own.assign(s.nwaves * SLICE, 0);
    s.seg.assign(s.nwaves * SLICE, 1);
    s.woff.assign(s.nwaves + 1, 0);
    s.gextra.assign(ng, 0);
    const bool timing = getenv("FCVM_PATTERN_TIMING") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "fcvm symmetric form: %-24s %.3f s\n", what, std::chrono::duration<double>(t - tprev).count());
        tprev = t;
    };
    const bool flagged_first = !(getenv("FCVM_SYM_FLAGGED_FIRST") && atoi(getenv("FCVM_SYM_FLAGGED_FIRST")) == 0);
    const std::vector<int32_t> grp = window_group_of(p, w);
    lap("group of every node");
    std::vector<uint16_t> &jfirst = s.jfirst;               // first kept block of the segment (index into the row's kept list)
    jfirst.assign(s.nwaves * SLICE, 0);
    s.rowtot.assign(s.nwaves * SLICE, 0);
    s.nextseg.assign(s.nwaves * SLICE, -1);
    std::vector<int64_t> drop(nthreads > 0 ? nthreads : 1, 0);
    struct VRow { int64_t len; int32_t nd; int32_t j0; uint32_t seg; int32_t tt; int32_t tot; };
    // kept blocks of row i of group g, in column order: calls f(block number in the row, column node, transposed product wanted)
    auto for_kept = [&](int32_t i, int32_t g, auto &&f) {
        for (int64_t q = p.rowptr[i]; q < p.rowptr[i + 1]; q++) {
            const int32_t j = p.colidx[q];
            if (j != i && grp[j] == g) {
                if (sym_keeps(i, j)) f((int)(q - p.rowptr[i]), j, true);
            } else {
                f((int)(q - p.rowptr[i]), j, false);
            }
        }
    };
    // transposed products every row receives (the kernels' 64-bit fixed-point sums are sized for SYM_MAX_RECEIVED of them):
    // recv[j] is written by the thread that owns j's group only -- both nodes of a counted pair are in that group
    std::vector<uint16_t> recv(nn, 0);
    // pass 1: new lengths, segments, lanes, entries per wave
    parallel_for(ng, nthreads, [&](int64_t b, int64_t e, int t) {
        std::vector<VRow> rows;
        std::vector<std::pair<int64_t, int64_t>> chain;
        for (int64_t g = b; g < e; g++) {
            rows.clear();
            int32_t extra = 0;
            for (int64_t at = g * R; at < (g + 1) * R; at++) {
                const int32_t nd = w.rownode[at];
                if (nd < 0 || (w.seg[at] & 255u) == 0) continue;
                for_kept(nd, (int32_t)g, [&](int, int32_t col, bool tr) { if (tr) recv[col]++; });
            }
            for (int64_t at = g * R; at < (g + 1) * R; at++) {       // the nodes of the group in the order of the full format's lanes
                const int32_t nd = w.rownode[at];
                if (nd < 0 || (w.seg[at] & 255u) == 0) continue;
                int64_t len = 0;
                for_kept(nd, (int32_t)g, [&](int, int32_t, bool) { len++; });
                drop[t] += (p.rowptr[nd + 1] - p.rowptr[nd]) - len;
                const int k = (int)std::max<int64_t>(1, (len + split - 1) / split);
                int64_t j0 = 0;
                for (int tt = 0; tt < k; tt++) {
                    const int64_t l = len / k + (tt < len % k ? 1 : 0);
                    const uint32_t code = tt == 0 ? ((uint32_t)extra << 8) | (uint32_t)k : ((uint32_t)(extra + tt - 1) << 8);
                    rows.push_back({l, nd, (int32_t)j0, code, tt, (int32_t)len});
                    j0 += l;
                }
                extra += k - 1;
            }
            s.gextra[g] = extra;
            // sorted over the whole group (uniform lengths inside a wave).  Sorting inside each wave only would keep a wave's rows
            // on one or two sliced-ELL slices and the copy kernel at 2.6 instead of 3.3 ms per assembly, but the 16-bit passes
            // lose 4 % to the ragged block rows (gpurun_out/r03bi, r03bj): the passes run 80 times per Newton iteration
            // Round 5 experiment (FCVM_SYM_CLUSTER_CUT=1): inside one length the uncut rows come first, then the first segments of cut rows, then their later segments (the kernels
            // need the lengths descending, nothing more): runs of eight consecutive lanes -- one 128-byte line of every value plane -- then hold
            // rows of ONE kind, which is what lets the assembly's work items be made of whole lane octets (wsym_asm_setup).  FCVM_SYM_CLUSTER_CUT=0:
            // the order of rounds 3-4 (cut segments where the full format's lane order left them).
            // mode 2: the kinds are dealt to the group's waves one after the other (uncut rows, first segments, later segments; lengths descending
            // inside a kind) and only THEN every wave is sorted by length (the prefix layout wants that per wave, not per group): all but two waves
            // of a group hold one kind.
            // Default 0 (rounds 3-4): mode 1 changes no time, mode 2 costs the SpMV 3.5 % (ragged block rows in the seam waves) -- profiles/r05_assembly_items.txt.
            static const int cluster_cut = getenv("FCVM_SYM_CLUSTER_CUT") ? atoi(getenv("FCVM_SYM_CLUSTER_CUT")) : 0;
            auto kind = [](const VRow &r) { return (r.seg & 255u) == 1u ? 0 : (r.tt == 0 ? 1 : 2); };
            if (cluster_cut == 2) {
                std::stable_sort(rows.begin(), rows.end(), [&](const VRow &x, const VRow &y) { return kind(x) < kind(y) || (kind(x) == kind(y) && x.len > y.len); });
                for (size_t k0 = 0; k0 < rows.size(); k0 += SLICE)
                    std::stable_sort(rows.begin() + k0, rows.begin() + std::min(rows.size(), k0 + (size_t)SLICE), [](const VRow &x, const VRow &y) { return x.len > y.len; });
            } else if (cluster_cut)
                std::stable_sort(rows.begin(), rows.end(), [&](const VRow &x, const VRow &y) { return x.len > y.len || (x.len == y.len && kind(x) < kind(y)); });
            else
                std::stable_sort(rows.begin(), rows.end(), [](const VRow &x, const VRow &y) { return x.len > y.len; });
            chain.clear();
            for (size_t k = 0; k < rows.size(); k++) {
                const int64_t wave = g * WV + (int64_t)k / SLICE, at = wave * SLICE + (int64_t)k % SLICE;
                s.rownode[at] = rows[k].nd;
                s.srcslot[at] = p.slotof[rows[k].nd];
                s.rowlen[at] = (uint16_t)rows[k].len;
                jfirst[at] = (uint16_t)rows[k].j0;
                s.seg[at] = rows[k].seg;
                s.woff[wave + 1] += rows[k].len;
                if (rows[k].tt == 0) s.rowtot[at] = (uint8_t)rows[k].tot;
                if ((rows[k].seg & 255u) != 1u) chain.push_back({((int64_t)rows[k].nd << 8) | rows[k].tt, at});   // a cut row
            }
            // the segments of a cut row, linked in order (the assembly owns whole rows and writes every segment)
            std::sort(chain.begin(), chain.end());
            for (size_t k = 0; k + 1 < chain.size(); k++)
                if ((chain[k].first >> 8) == (chain[k + 1].first >> 8)) s.nextseg[chain[k].second] = (int32_t)chain[k + 1].second;
        }
    }, 16);
    lap("pass 1 (lengths, lanes)");
    s.dropped = 0;
    for (int64_t v : drop) s.dropped += v;
    s.trmax = 0;
    for (int64_t i = 0; i < nn; i++) s.trmax = std::max<int64_t>(s.trmax, recv[i]);
    if (s.trmax > SYM_MAX_RECEIVED) return false;      // (the full form stays in charge)
    s.xmax = 0;
    for (int64_t g = 0; g < ng; g++) s.xmax = std::max<int64_t>(s.xmax, s.gextra[g]);
    for (int64_t i = 0; i < s.nwaves; i++) s.woff[i + 1] = (s.woff[i + 1] + 15) / 16 * 16 + s.woff[i];
    s.nblocks = s.woff[s.nwaves];
    s.cols.resize(s.nblocks);
    s.jell.resize(s.nblocks);
    s.kmap.resize(p.nnzb);
    s.diagk.assign(nn, 0);
    s.maxl = 0;
    for (int64_t i = 0; i < s.nwaves; i++) s.maxl = std::max<int>(s.maxl, s.rowlen[i * SLICE]);
    s.rowoff.assign(s.nwaves * (s.maxl + 1), 0);
    lap("offsets, allocation");
    // pass 2: entries in the prefix layout
    parallel_for(ng, nthreads, [&](int64_t b, int64_t e, int) {
        std::vector<uint16_t> pos(nn, 0), slot(nn, 0);     // window position / first-segment position of a node of the current group
        for (int64_t g = b; g < e; g++) {
            const int32_t *wl = &w.wlist[w.gptr[g]];
            const int64_t W = w.gptr[g + 1] - w.gptr[g];
            for (int64_t i = 0; i < W; i++) pos[wl[i]] = (uint16_t)i;
            for (int64_t at = g * R; at < (g + 1) * R; at++)
                if (s.rownode[at] >= 0 && (s.seg[at] & 255u) > 0) {
                    const int32_t nd = s.rownode[at];
                    slot[nd] = (uint16_t)(at - g * R);
                    std::fill(s.kmap.begin() + p.rowptr[nd], s.kmap.begin() + p.rowptr[nd + 1], (uint8_t)0xFF);
                }
            for (int64_t wave = g * WV; wave < (g + 1) * WV; wave++) {
                // pad entries of the wave (alignment) are never read as blocks; fill them harmlessly
                for (int64_t q = s.woff[wave]; q < s.woff[wave + 1]; q++) { s.cols[q] = 0xFFFF0000u; s.jell[q] = 0; }
                const uint16_t *len = &s.rowlen[wave * SLICE];
                // first entry of block row j of the wave: lanes are sorted by length, block row j holds the lanes with len > j
                int64_t rowoff[256];
                {
                    int64_t off = s.woff[wave];
                    int n = SLICE;
                    for (int64_t jj = 0; jj < len[0]; jj++) {
                        while (n > 0 && len[n - 1] <= jj) n--;
                        rowoff[jj] = off;
                        s.rowoff[wave * (s.maxl + 1) + jj] = (uint16_t)(off - s.woff[wave]);
                        off += n;
                    }
                    s.rowoff[wave * (s.maxl + 1) + len[0]] = (uint16_t)(off - s.woff[wave]);
                }
                for (int64_t l = 0; l < SLICE; l++) {
                    const int32_t nd = s.rownode[wave * SLICE + l];
                    if (nd < 0) continue;
                    s.own[wave * SLICE + l] = pos[nd];
                    // the row's kept blocks with the ones that want a transposed product FIRST: beyond the largest such count
                    // of a wave no lane has one and the wave skips that code altogether (in column order every block row had some)
                    const int64_t j0 = jfirst[wave * SLICE + l], j1 = j0 + len[l];
                    int64_t nflag = 0;
                    for_kept(nd, (int32_t)g, [&](int, int32_t, bool tr) { nflag += tr ? 1 : 0; });
                    if (!flagged_first) nflag = -1;
                    int64_t kf = 0, ku = nflag, kall = 0;
                    for_kept(nd, (int32_t)g, [&](int jq, int32_t col, bool tr) {
                        const int64_t k = nflag < 0 ? kall++ : (tr ? kf++ : ku++);
                        if (k >= j0 && k < j1) {
                            s.cols[rowoff[k - j0] + l] = (uint32_t)pos[col] | ((tr ? (uint32_t)slot[col] : 0xFFFFu) << 16);
                            s.jell[rowoff[k - j0] + l] = (uint8_t)jq;
                            s.kmap[p.rowptr[nd] + jq] = (uint8_t)k;
                            if (col == nd) s.diagk[nd] = (uint8_t)k;
                        }
                    });
                }
            }
        }
    }, 16);
    lap("pass 2 (entries)");
    if (getenv("FCVM_SYM_STATS")) {      // what the assembly's work items (16 lanes of a wave) look like: kept blocks of their rows
        int64_t hist[9] = {0}, items = 0, sum_max = 0, sum_tot = 0, first = 0, later = 0, empty = 0;
        const int lim[9] = {14, 20, 28, 34, 40, 46, 53, 64, 255};
        for (int64_t it = 0; it < s.nwaves * 4; it++) {
            int m = 0, nf = 0;
            for (int l = 0; l < 16; l++) {
                const int64_t at = it * 16 + l;
                if (s.rownode[at] < 0) { empty++; continue; }
                if ((s.seg[at] & 255u) == 0) { later++; continue; }
                first++; nf++;
                m = std::max<int>(m, s.rowtot[at]);
                sum_tot += s.rowtot[at];
            }
            if (!nf) continue;
            items++;
            sum_max += 16 * m;
            int b = 0;
            while (m > lim[b]) b++;
            hist[b]++;
        }
        fprintf(stderr, "fcvm symmetric form: %lld items; lanes: %lld rows, %lld later segments, %lld empty; kept blocks %lld, 16 x item maximum %lld (%.2f x)\n   items by longest row:",
                (long long)items, (long long)first, (long long)later, (long long)empty, (long long)sum_tot, (long long)sum_max, (double)sum_max / (double)sum_tot);
        for (int b = 0; b < 9; b++) fprintf(stderr, " <=%d: %lld", lim[b], (long long)hist[b]);
        fprintf(stderr, "\n");
    }
    return true;
}

int64_t check_window_sym(const Pattern &p, const WinPattern &w, const WinSym &s) {
    int64_t bad = 0;
    const int64_t R = (int64_t)s.waves * SLICE;
    std::vector<uint8_t> hit(p.nnzb, 0);
    std::vector<int32_t> recv(p.nn, 0);      // transposed products per receiving row: bounded, see SYM_MAX_RECEIVED
    auto mark = [&](int32_t i, int32_t j) {
        const int32_t *rb = p.colidx.data() + p.rowptr[i], *re = p.colidx.data() + p.rowptr[i + 1];
        const int32_t *at = std::lower_bound(rb, re, j);
        if (at == re || *at != j) { bad++; return; }
        if (hit[at - p.colidx.data()]++) bad++;
    };
    for (int64_t wave = 0; wave < s.nwaves; wave++) {
        const int64_t g = wave / s.waves;
        const int32_t *wl = &w.wlist[w.gptr[g]];
        const int64_t W = w.gptr[g + 1] - w.gptr[g];
        const uint16_t *len = &s.rowlen[wave * SLICE];
        for (int l = 0; l < SLICE; l++) {
            const int64_t at = wave * SLICE + l;
            if (l > 0 && len[l] > len[l - 1]) bad++;
            if (s.rownode[at] < 0) { if (len[l]) bad++; continue; }
            if (len[l] > s.split) bad++;
            if (s.own[at] >= W || wl[s.own[at]] != s.rownode[at]) bad++;
            if (s.srcslot[at] != p.slotof[s.rownode[at]]) bad++;
        }
        int64_t off = s.woff[wave];
        for (int64_t j = 0; j < len[0]; j++) {
            int64_t n = 0;
            while (n < SLICE && len[n] > j) n++;
            for (int64_t l = 0; l < n; l++) {
                const int32_t i = s.rownode[wave * SLICE + l];
                const uint32_t en = s.cols[off + l];
                const uint32_t cp = en & 0xFFFFu, ts = en >> 16;
                if ((int64_t)cp >= W) { bad++; continue; }
                const int32_t jn = wl[cp];
                mark(i, jn);
                if (p.colidx[p.rowptr[i] + s.jell[off + l]] != jn) bad++;           // the copy kernel's source is this block
                // the assembly's tables: kept number of the block, position of the block row in the wave
                if (s.kmap[p.rowptr[i] + s.jell[off + l]] != (uint8_t)(s.jfirst[wave * SLICE + l] + j)) bad++;
                if (l == 0 && s.rowoff[wave * (s.maxl + 1) + j] != (uint16_t)(off - s.woff[wave])) bad++;
                if (ts != 0xFFFFu) {
                    const int64_t tat = g * R + ts;
                    if ((int64_t)ts >= R || s.rownode[tat] != jn || (s.seg[tat] & 255u) == 0 || jn == i) bad++;
                    else { mark(jn, i); recv[jn]++; }
                }
            }
            off += n;
        }
        if (off > s.woff[wave + 1] || s.woff[wave] % 16) bad++;
    }
    for (int64_t q = 0; q < p.nnzb; q++)
        if (hit[q] != 1) bad++;
    // segment chains of cut rows: same node, kept numbers follow on, total = rowtot; diagonal blocks; blocks not stored
    std::vector<uint8_t> seen(p.nn, 0);
    for (int64_t at = 0; at < s.nwaves * SLICE; at++) {
        const int32_t i = s.rownode[at];
        if (i < 0 || (s.seg[at] & 255u) == 0) { if (s.rowtot[at]) bad++; continue; }
        if (seen[i]++) bad++;
        int64_t tot = 0, cur = at, nseg = 0;
        while (cur >= 0) {
            if (s.rownode[cur] != i || s.jfirst[cur] != tot || (cur != at && (s.seg[cur] & 255u) != 0)) bad++;
            if (cur / R != at / R) bad++;                       // all segments in the row's group
            tot += s.rowlen[cur];
            nseg++;
            cur = s.nextseg[cur];
            if (nseg > 256) { bad++; break; }
        }
        if (tot != s.rowtot[at] || nseg != (int64_t)(s.seg[at] & 255u)) bad++;
        int64_t kept = 0;
        for (int64_t q = p.rowptr[i]; q < p.rowptr[i + 1]; q++) {
            if (s.kmap[q] != 0xFF) kept++;
            if (p.colidx[q] == i && s.kmap[q] != s.diagk[i]) bad++;
        }
        if (kept != tot) bad++;
    }
    for (int64_t i = 0; i < p.nn; i++)
        if (!seen[i]) bad++;
    int64_t most = 0;
    for (int64_t i = 0; i < p.nn; i++) most = std::max<int64_t>(most, recv[i]);
    if (most != s.trmax || most > SYM_MAX_RECEIVED) bad++;
    return bad;
}

}  // namespace fcvm
