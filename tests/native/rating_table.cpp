// tests/native/rating_table.cpp -- host/nmi_rating.hpp against the text of the reference (no GPU): the six-level view aliases
// the flat table in find_max_elements' scan order (helperFunctions.cpp:53-64), and helperFunctions::find_max_elements --
// the reference's signature -- returns what the reference's two passes return (maximum from 0 with strict '>', then every
// cell EQUAL to it in scan order) on random tables, tables with ties, all-zero, all-negative and NaN tables, for a view of an
// NmiRatingTable and for a pointer tree allocated level by level the way localization.cpp:185-210 does.
// Built and run by tests/test_rating_table.py (also under ASan / UBSan).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "nmi_rating.hpp"

static unsigned rng_state = 777u;
static unsigned rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

struct Cell {
    int sx, sy, sz, wx, wy, wz;
    float v;
};

// the reference's two passes, spelled out as in helperFunctions.cpp:50-103 (the test's own restatement, the checker)
static std::vector<Cell> brute(float ******nmi, NmiSearchKernel &k)
{
    float max = 0;
    for (int wz = 0; wz < k.getNumWarpZ(); wz++)
        for (int wy = 0; wy < k.getNumWarpY(); wy++)
            for (int wx = 0; wx < k.getNumWarpX(); wx++)
                for (int sz = 0; sz < k.getNumSynthZ(); sz++)
                    for (int sy = 0; sy < k.getNumSynthY(); sy++)
                        for (int sx = 0; sx < k.getNumSynthX(); sx++)
                            if (nmi[wz][wy][wx][sz][sy][sx] > max) max = nmi[wz][wy][wx][sz][sy][sx];
    std::vector<Cell> out;
    for (int wz = 0; wz < k.getNumWarpZ(); wz++)
        for (int wy = 0; wy < k.getNumWarpY(); wy++)
            for (int wx = 0; wx < k.getNumWarpX(); wx++)
                for (int sz = 0; sz < k.getNumSynthZ(); sz++)
                    for (int sy = 0; sy < k.getNumSynthY(); sy++)
                        for (int sx = 0; sx < k.getNumSynthX(); sx++)
                            if (nmi[wz][wy][wx][sz][sy][sx] == max) out.push_back({sx, sy, sz, wx, wy, wz, nmi[wz][wy][wx][sz][sy][sx]});
    return out;
}

static int check(float ******nmi, NmiSearchKernel &k, const char *what)
{
    std::vector<Cell> exp = brute(nmi, k);
    std::vector<NmiSearchKernel> got = helperFunctions::find_max_elements(nmi, k);
    if (got.size() != exp.size()) {
        printf("%s: %zu ties, expected %zu\n", what, got.size(), exp.size());
        return 1;
    }
    for (size_t i = 0; i < exp.size(); ++i) {
        NmiSearchKernel &g = got[i];
        const Cell &e = exp[i];
        if (g.getBestSynthX() != e.sx || g.getBestSynthY() != e.sy || g.getBestSynthZ() != e.sz || g.getBestWarpX() != e.wx ||
            g.getBestWarpY() != e.wy || g.getBestWarpZ() != e.wz || !(g.getNmi() == e.v)) {
            printf("%s: tie %zu differs\n", what, i);
            return 1;
        }
        // a tie carries indices and score only: the counts and steps stay at the blank state (-1), as `NmiSearchKernel()` leaves them
        if (g.getNumSynthX() != -1 || g.getStepRadZ() != -1) {
            printf("%s: tie %zu is not a blank descriptor\n", what, i);
            return 1;
        }
    }
    return 0;
}

int main()
{
    const int shapes[][6] = {{3, 3, 3, 3, 3, 3}, {1, 1, 1, 1, 1, 1}, {3, 1, 2, 1, 4, 1}, {5, 2, 1, 2, 2, 3}, {1, 1, 1, 3, 3, 3}, {4, 4, 4, 1, 1, 1}};
    for (const auto &sh : shapes) {
        NmiSearchKernel k(sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
        NmiRatingTable table(k);
        float ******rating = table;  // reads like NmiObjects::rating
        const long long S = table.numSynths(), Wn = table.numWarps();
        if (table.size() != S * Wn) return 2;
        // 1. the view aliases the flat table in scan order: ratings[w * S + s]
        for (long long i = 0; i < table.size(); ++i) table.flat()[i] = (float)i;
        long long scan = 0;
        for (int wz = 0; wz < sh[5]; wz++)
            for (int wy = 0; wy < sh[4]; wy++)
                for (int wx = 0; wx < sh[3]; wx++)
                    for (int sz = 0; sz < sh[2]; sz++)
                        for (int sy = 0; sy < sh[1]; sy++)
                            for (int sx = 0; sx < sh[0]; sx++, scan++) {
                                const long long w = ((long long)wz * sh[4] + wy) * sh[3] + wx, s = ((long long)sz * sh[1] + sy) * sh[0] + sx;
                                if (rating[wz][wy][wx][sz][sy][sx] != (float)scan || scan != w * S + s) return 3;
                                if (&rating[wz][wy][wx][sz][sy][sx] != table.flat() + scan) return 4;
                            }
        if (check(rating, k, "ramp")) return 5;  // unique maximum: the last cell
        // 2. random tables: smooth values, few distinct values (many ties), negative, NaN, all zero
        for (int rep = 0; rep < 200; ++rep) {
            const int kind = rep % 5;
            for (long long i = 0; i < table.size(); ++i) {
                float v;
                if (kind == 0) v = (rnd() % 100000) / 100000.0f;
                else if (kind == 1) v = (float)(rnd() % 3) * 0.25f;
                else if (kind == 2) v = -(float)(rnd() % 7) - 1.0f;
                else if (kind == 3) v = (rnd() % 4 == 0) ? NAN : (float)(rnd() % 5) * 0.1f - 0.2f;
                else v = 0.0f;
                table.flat()[i] = v;
            }
            if (kind == 2 && rep % 10 == 2) table.flat()[rnd() % table.size()] = 0.0f;  // one zero among negatives: it is THE tie
            if (check(rating, k, "random")) return 6;
            // the flat function and the six-level one agree on the count and on the first winner
            float mx = 0;
            const long long n = nmi_find_max_elements(table.flat(), table.size(), nullptr, 0, &mx);
            std::vector<NmiSearchKernel> got = helperFunctions::find_max_elements(rating, k);
            if ((long long)got.size() != n) return 7;
            if (n > 0 && !(got[0].getNmi() == mx)) return 8;
        }
        // 3. a tree allocated level by level, as localization.cpp:185-210 does (the function takes ANY tree of the shape)
        float ******ref = new float *****[sh[5]];
        for (int wz = 0; wz < sh[5]; wz++) {
            ref[wz] = new float ****[sh[4]];
            for (int wy = 0; wy < sh[4]; wy++) {
                ref[wz][wy] = new float ***[sh[3]];
                for (int wx = 0; wx < sh[3]; wx++) {
                    ref[wz][wy][wx] = new float **[sh[2]];
                    for (int sz = 0; sz < sh[2]; sz++) {
                        ref[wz][wy][wx][sz] = new float *[sh[1]];
                        for (int sy = 0; sy < sh[1]; sy++) {
                            ref[wz][wy][wx][sz][sy] = new float[sh[0]];
                            for (int sx = 0; sx < sh[0]; sx++) ref[wz][wy][wx][sz][sy][sx] = (float)(rnd() % 4) * 0.3f;
                        }
                    }
                }
            }
        }
        const int bad = check(ref, k, "level-by-level tree");
        for (int wz = 0; wz < sh[5]; wz++) {
            for (int wy = 0; wy < sh[4]; wy++) {
                for (int wx = 0; wx < sh[3]; wx++) {
                    for (int sz = 0; sz < sh[2]; sz++) {
                        for (int sy = 0; sy < sh[1]; sy++) delete[] ref[wz][wy][wx][sz][sy];
                        delete[] ref[wz][wy][wx][sz];
                    }
                    delete[] ref[wz][wy][wx];
                }
                delete[] ref[wz][wy];
            }
            delete[] ref[wz];
        }
        delete[] ref;
        if (bad) return 9;
        // 4. resize re-points the view (NMIobjectsReInitialization, localization.cpp:403-420)
        table.resize(2, 1, 1, 1, 1, 2);
        float ******r2 = table.view();
        r2[1][0][0][0][0][1] = 0.5f;
        if (table.size() != 4 || table.flat()[3] != 0.5f) return 10;
    }
    // an empty grid (the blank descriptor's counts are -1): no ties, no access
    NmiSearchKernel blank;
    NmiRatingTable none(blank);
    if (!helperFunctions::find_max_elements(none.view(), blank).empty() || none.size() != 0) return 11;
    printf("rating table ok\n");
    return 0;
}
