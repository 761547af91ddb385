#include "fcvm_pattern.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
using namespace fcvm;
// structured tet10-like mesh: cells of a lattice split into 6 tets; mid-side nodes via a map
#include <map>
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<double> xyz;
    std::map<std::pair<int,int>, int> mid;
    auto vid = [&](int i, int j, int k) { return (i * (n + 1) + j) * (n + 1) + k; };
    for (int i = 0; i <= n; i++) for (int j = 0; j <= n; j++) for (int k = 0; k <= n; k++) { xyz.push_back(i * 2.0); xyz.push_back(j); xyz.push_back(k * 0.1); }
    std::vector<int32_t> el;
    const int T[6][4] = {{0,1,3,7},{0,3,2,7},{0,2,6,7},{0,6,4,7},{0,4,5,7},{0,5,1,7}};
    const int MP[6][2] = {{0,1},{1,2},{0,2},{0,3},{1,3},{2,3}};
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {
        int c[8]; for (int b = 0; b < 8; b++) c[b] = vid(i + ((b >> 2) & 1), j + ((b >> 1) & 1), k + (b & 1));
        for (int t = 0; t < 6; t++) {
            int v[4]; for (int a = 0; a < 4; a++) v[a] = c[T[t][a]];
            for (int a = 0; a < 4; a++) el.push_back(v[a]);
            for (int m = 0; m < 6; m++) {
                int a = v[MP[m][0]], b = v[MP[m][1]]; auto key = std::make_pair(std::min(a,b), std::max(a,b));
                auto it = mid.find(key);
                if (it == mid.end()) { int id = (int)(xyz.size() / 3); for (int q = 0; q < 3; q++) xyz.push_back(0.5 * (xyz[3*a+q] + xyz[3*b+q])); it = mid.insert({key, id}).first; }
                el.push_back(it->second);
            }
        }
    }
    const int64_t ne = (int64_t)el.size() / 10, nn = (int64_t)xyz.size() / 3;
    Pattern p; WinPattern w; WinSym s;
    build_pattern(ne, nn, el.data(), p, 4, SORT_WINDOW, 10, nullptr, true);
    std::vector<int32_t> order = morton_order(nn, xyz.data());
    Pattern p2; build_pattern(ne, nn, el.data(), p2, 4, SORT_WINDOW, 10, order.data(), false);
    bool ok = build_window_format(p2, xyz.data(), 4, 65535, w);
    bool ok2 = ok && build_window_sym(p2, w, 4, s);
    long long bad = ok2 ? (long long)check_window_sym(p2, w, s) : -1;
    printf("ne %lld nn %lld window %d sym %d bad %lld stored %lld of %lld\n", (long long)ne, (long long)nn, (int)ok, (int)ok2, bad, (long long)s.nblocks, (long long)w.nblocks);
    return bad != 0;
}
