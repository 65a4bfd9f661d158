// Unit test of the C++ class NmiSearchKernel (host/nmi_search_kernel.hpp) against the behaviour of the reference class
// (Thirdparty/Localization/nmiSearchKernel.cpp:25-195), plain g++, no GPU.  Returns 0 when every check holds.
#include <cmath>
#include <cstdio>
#include <sstream>
#include <string>

#include "nmi_search_kernel.hpp"

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            printf("CHECK failed at line %d: %s\n", __LINE__, #cond);      \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

int main()
{
    NmiSearchKernel blank;  // default ctor: -1 everywhere, NMI 0 (nmiSearchKernel.cpp:35-38)
    CHECK(blank.numSynthX == -1 && blank.numWarpZ == -1 && blank.stepX == -1.0f && blank.stepRadZ == -1.0f);
    CHECK(blank.bestSynthX == -1 && blank.bestWarpZ == -1 && blank.NMI == 0.0f);

    NmiSearchKernel k(3, 3, 5, 3, 1, 3, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);  // 12-argument ctor (:25-33)
    CHECK(k.getNumSynthZ() == 5 && k.getNumWarpY() == 1 && k.getStepZ() == 0.5f && k.getStepRadX() == 0.02f);
    CHECK(k.getBestSynthX() == -1 && k.getNmi() == 0.0f);

    k.setBest(1, 1, 2, 1, 0, 1, 0.42f);
    CHECK(k.isMiddle());  // 1==3/2, 1==3/2, 2==5/2, 1==3/2, 0==1/2, 1==3/2
    k.setBest(1, 1, 3, 1, 0, 1, 0.42f);
    CHECK(!k.isMiddle());

    NmiSearchKernel other;
    other.setBest(&k);  // indices only: NMI is not copied by this overload (:83-91)
    CHECK(other.bestSynthZ == 3 && other.NMI == 0.0f && other.numSynthX == -1);
    other.setKernel(&k);
    CHECK(other.numSynthZ == 5 && other.stepRadZ == 0.05f && other.NMI == 0.0f);
    NmiSearchKernel copy;
    copy.setTo(&k);  // kernel + best + NMI (:93-98)
    CHECK(copy.NMI == 0.42f && copy.bestSynthZ == 3 && copy.numWarpY == 1);

    // resizeKernel (:104-141): z-synth best (3) is interior of 5 -> halves; x-synth best 1 interior -> halves;
    // warpY has a single cell -> always halves; nothing falls under the minimum here
    k.resizeKernel();
    CHECK(std::fabs(k.stepX - 0.1f) < 1e-7f && std::fabs(k.stepZ - 0.25f) < 1e-7f && std::fabs(k.stepRadY - 0.01f) < 1e-7f);
    CHECK(k.numSynthX == 3 && k.numWarpY == 1);
    // border best keeps its step; a step under 0.005 m / 0.001 rad collapses the axis
    NmiSearchKernel b(3, 3, 3, 3, 3, 3, 0.008f, 0.2f, 0.5f, 0.0015f, 0.02f, 0.05f);
    b.setBest(1, 0, 2, 1, 2, 1, 0.1f);
    b.resizeKernel();
    CHECK(b.numSynthX == 1 && b.stepX == 0.004f && b.stepY == 0.2f && b.stepZ == 0.5f);
    CHECK(b.numWarpX == 1 && b.stepRadY == 0.02f && std::fabs(b.stepRadZ - 0.025f) < 1e-7f);

    // operator<< (:183-195)
    NmiSearchKernel p(3, 3, 3, 3, 3, 3, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
    p.setBest(1, 0, 2, 1, 1, 1, 0.28606f);
    std::ostringstream os;
    os << p;
    const std::string exp =
        "sX:  1/3: 0.20000;\t sY:  0/3: 0.20000;\t sZ:  2/3: 0.50000;\t rX:  1/3: 0.02000;\t rY:  1/3: 0.02000;\t rZ:  1/3: 0.05000;\t NMI: 0.28606";
    CHECK(os.str() == exp);

    p.resetBest();
    CHECK(p.bestWarpX == -1 && p.NMI == 0.0f && p.numSynthX == 3);
    p.resetKernel();
    CHECK(p.numSynthX == -1 && p.stepRadZ == -1.0f);
    p.reset();
    CHECK(p.NMI == 0.0f && p.bestSynthX == -1);

    // round trip through the C struct
    nmi_search_kernel c = copy.to_c();
    NmiSearchKernel back(c);
    CHECK(back.numSynthZ == 5 && back.bestSynthZ == 3 && back.NMI == 0.42f && back.stepRadZ == 0.05f);

    printf(failures ? "%d FAILURES\n" : "search kernel class ok\n", failures);
    return failures ? 1 : 0;
}
