// nmi_rating.hpp -- the reference's rating container and its arg-max, with the reference's own signatures (header only, over the
// C ABI of include/nmi_host.h), so that the lines of Tracking::RelocalizeWithNMI that USE the table compile unchanged:
//     nmiObj->rating[wZ][wY][wX][sZ][sY][sX] = nmi;                                          (src/Tracking.cc:1895)
//     maxElements = helperFunctions::find_max_elements(nmiObj->rating, *nmiObj->NmiKernel);   (src/Tracking.cc:1905)
//
//   NmiRatingTable   NmiObjects::rating (Thirdparty/Localization/localization.hpp:36; allocated at localization.cpp:185-210
//                    and again on every grid resize, :256-350): a 6-level pointer tree [wZ][wY][wX][sZ][sY][sX] of float.  Here
//                    the cells are ONE flat array in the scan order of find_max_elements -- which is the layout of the rating
//                    table nmi_search_grid writes, ratings[w * S + s] with w = (wz * nWy + wy) * nWx + wx and
//                    s = (sz * nSy + sy) * nSx + sx -- and the five pointer levels are views into it: a device table is copied
//                    into flat() with one hipMemcpy and read through view() as the reference reads its own.
//   helperFunctions::find_max_elements(float ******, NmiSearchKernel &)
//                    Thirdparty/Localization/helperFunctions.cpp:50-103: the maximum starts at 0, strict '>', then every cell
//                    EQUAL to it is returned in scan order (all-zero table: every cell; all negative / NaN: none).  Takes any
//                    pointer tree of that shape, the reference's own included.
#pragma once
#include <stdint.h>

#include <vector>

#include "nmi_host.h"
#include "nmi_search_kernel.hpp"

class NmiRatingTable {
public:
    NmiRatingTable() = default;
    NmiRatingTable(int numSynthX, int numSynthY, int numSynthZ, int numWarpX, int numWarpY, int numWarpZ)
    {
        resize(numSynthX, numSynthY, numSynthZ, numWarpX, numWarpY, numWarpZ);
    }
    explicit NmiRatingTable(NmiSearchKernel &k)
    {
        resize(k.getNumSynthX(), k.getNumSynthY(), k.getNumSynthZ(), k.getNumWarpX(), k.getNumWarpY(), k.getNumWarpZ());
    }
    NmiRatingTable(const NmiRatingTable &) = delete;  // (the pointer levels point into this object's own cells)
    NmiRatingTable &operator=(const NmiRatingTable &) = delete;

    // NMIobjectsReInitialization / setNmiObjectsKernel (localization.cpp:390-420) re-allocate the tree for a new grid; cells are zeroed
    void resize(int numSynthX, int numSynthY, int numSynthZ, int numWarpX, int numWarpY, int numWarpZ)
    {
        n_[0] = numSynthX, n_[1] = numSynthY, n_[2] = numSynthZ, n_[3] = numWarpX, n_[4] = numWarpY, n_[5] = numWarpZ;
        for (int &v : n_)
            if (v < 0) v = 0;
        const size_t sx = (size_t)n_[0], sy = (size_t)n_[1], sz = (size_t)n_[2], wx = (size_t)n_[3], wy = (size_t)n_[4], wz = (size_t)n_[5];
        cells_.assign(wz * wy * wx * sz * sy * sx, 0.0f);
        p1_.resize(wz * wy * wx * sz * sy);
        p2_.resize(wz * wy * wx * sz);
        p3_.resize(wz * wy * wx);
        p4_.resize(wz * wy);
        p5_.resize(wz);
        for (size_t i = 0; i < p1_.size(); ++i) p1_[i] = cells_.data() + i * sx;
        for (size_t i = 0; i < p2_.size(); ++i) p2_[i] = p1_.data() + i * sy;
        for (size_t i = 0; i < p3_.size(); ++i) p3_[i] = p2_.data() + i * sz;
        for (size_t i = 0; i < p4_.size(); ++i) p4_[i] = p3_.data() + i * wx;
        for (size_t i = 0; i < p5_.size(); ++i) p5_[i] = p4_.data() + i * wy;
    }
    void resize(NmiSearchKernel &k) { resize(k.getNumSynthX(), k.getNumSynthY(), k.getNumSynthZ(), k.getNumWarpX(), k.getNumWarpY(), k.getNumWarpZ()); }

    float ******view() { return p5_.data(); }           // rating[wZ][wY][wX][sZ][sY][sX]
    operator float ******() { return p5_.data(); }       // so that an NmiRatingTable member named `rating` reads like the reference's
    float *flat() { return cells_.data(); }              // [numWarps() * numSynths()], index w * numSynths() + s: nmi_search_grid's table
    const float *flat() const { return cells_.data(); }
    int64_t numSynths() const { return (int64_t)n_[0] * n_[1] * n_[2]; }
    int64_t numWarps() const { return (int64_t)n_[3] * n_[4] * n_[5]; }
    int64_t size() const { return (int64_t)cells_.size(); }

private:
    int n_[6] = {0, 0, 0, 0, 0, 0};
    std::vector<float> cells_;
    std::vector<float *> p1_;
    std::vector<float **> p2_;
    std::vector<float ***> p3_;
    std::vector<float ****> p4_;
    std::vector<float *****> p5_;
};

namespace helperFunctions {

inline std::vector<NmiSearchKernel> find_max_elements(float ******nmi, NmiSearchKernel &nmiKernel)
{
    const int nsx = nmiKernel.getNumSynthX(), nsy = nmiKernel.getNumSynthY(), nsz = nmiKernel.getNumSynthZ();
    const int nwx = nmiKernel.getNumWarpX(), nwy = nmiKernel.getNumWarpY(), nwz = nmiKernel.getNumWarpZ();
    std::vector<float> flat;
    if (nsx > 0 && nsy > 0 && nsz > 0 && nwx > 0 && nwy > 0 && nwz > 0) flat.reserve((size_t)nsx * nsy * nsz * nwx * nwy * nwz);
    for (int wz = 0; wz < nwz; wz++)  // the scan order of helperFunctions.cpp:53-64
        for (int wy = 0; wy < nwy; wy++)
            for (int wx = 0; wx < nwx; wx++)
                for (int sz = 0; sz < nsz; sz++)
                    for (int sy = 0; sy < nsy; sy++)
                        for (int sx = 0; sx < nsx; sx++) flat.push_back(nmi[wz][wy][wx][sz][sy][sx]);
    std::vector<NmiSearchKernel> maxElements;
    if (flat.empty()) return maxElements;
    float max_value = 0.0f;
    const int64_t ties = nmi_find_max_elements(flat.data(), (int64_t)flat.size(), nullptr, 0, &max_value);
    if (ties <= 0) return maxElements;
    std::vector<int64_t> idx((size_t)ties);
    nmi_find_max_elements(flat.data(), (int64_t)flat.size(), idx.data(), ties, &max_value);
    maxElements.reserve((size_t)ties);
    for (int64_t t : idx) {
        int64_t r = t;
        const int sx = (int)(r % nsx); r /= nsx;
        const int sy = (int)(r % nsy); r /= nsy;
        const int sz = (int)(r % nsz); r /= nsz;
        const int wx = (int)(r % nwx); r /= nwx;
        const int wy = (int)(r % nwy); r /= nwy;
        const int wz = (int)r;
        NmiSearchKernel maxElem = NmiSearchKernel();
        maxElem.setBest(sx, sy, sz, wx, wy, wz, nmi[wz][wy][wx][sz][sy][sx]);
        maxElements.push_back(maxElem);
    }
    return maxElements;
}

}  // namespace helperFunctions
