// ref_search_kernel_driver.cpp -- TEST INFRASTRUCTURE ONLY: drives the REFERENCE's own NmiSearchKernel class.
//
// Linked (oracle/Makefile, target _ref) against the object code of the unmodified reference file
//   /root/reference/Thirdparty/Localization/nmiSearchKernel.cpp  (+ allProperties.hpp, nmiSearchKernel.hpp)
// compiled where it lies; nothing of the reference is copied into this repository.  The only accommodation is an
// include-directory alias oracle/_ref/inc/NmiSearchKernel.hpp -> nmiSearchKernel.hpp, because the reference spells
// its own header with the Windows file system's case insensitivity (nmiSearchKernel.cpp:21).
//
// The binary stays in this container (oracle/_ref/ is git-ignored); tests/golden/make_search_kernel_ref.py runs it
// over seeded inputs and commits the answers as tests/golden/search_kernel_ref.npz, which is what pins SURVEY.md row
// a14 (isMiddle :99-102, resizeKernel :104-141, operator<< :183-195, constructors / setters / resets :25-158) to the
// reference's object code.
//
// Protocol (stdin -> stdout), one case per input line:
//   in : n0..n5  s0..s5(hex f32 bits)  b0..b5  nmi(hex f32 bits)  R
//   out: line 1  isMiddle  then for r = 0..R: six counts and six step bit patterns after r calls of resizeKernel()
//        line 2  operator<< of the initial state;  line 3  operator<< after the R resizes
// Mode "walk": a scripted sequence over every other public member, one state line per call (see walk()).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>

#include "NmiSearchKernel.hpp"  // the reference's header, through the case alias

static uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static float from_bits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static void state(NmiSearchKernel &k)
{
    printf(" %d %d %d %d %d %d", k.getNumSynthX(), k.getNumSynthY(), k.getNumSynthZ(), k.getNumWarpX(), k.getNumWarpY(),
           k.getNumWarpZ());
    printf(" %08x %08x %08x %08x %08x %08x", bits(k.getStepX()), bits(k.getStepY()), bits(k.getStepZ()), bits(k.getStepRadX()),
           bits(k.getStepRadY()), bits(k.getStepRadZ()));
}

static void full_state(const char *tag, NmiSearchKernel &k)
{
    printf("%s", tag);
    state(k);
    printf(" %d %d %d %d %d %d %08x\n", k.getBestSynthX(), k.getBestSynthY(), k.getBestSynthZ(), k.getBestWarpX(),
           k.getBestWarpY(), k.getBestWarpZ(), bits(k.getNmi()));
}

// Every public member that the per-case protocol does not reach, in a fixed script; the test replays the same script
// on orbslam2_nmi_amd/host/nmi_search_kernel.hpp and compares the lines.
static int walk()
{
    NmiSearchKernel blank;
    full_state("default_ctor", blank);
    NmiSearchKernel k(3, 3, 5, 3, 1, 3, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
    full_state("ctor12", k);
    k.setBest(1, 1, 3, 1, 0, 1, 0.42f);
    full_state("setBest7", k);
    NmiSearchKernel other;
    other.setBest(&k);
    full_state("setBest_ptr", other);
    other.setKernel(&k);
    full_state("setKernel_ptr", other);
    NmiSearchKernel copy;
    copy.setTo(&k);
    full_state("setTo", copy);
    copy.setKernel(2, 4, 6, 8, 10, 12, 1.5f, 2.5f, 3.5f, 0.125f, 0.25f, 0.375f);
    full_state("setKernel12", copy);
    copy.resetBest();
    full_state("resetBest", copy);
    copy.resetKernel();
    full_state("resetKernel", copy);
    k.reset();
    full_state("reset", k);
    std::ostringstream os;
    os << blank;
    printf("stream_default %s\n", os.str().c_str());
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "walk")) return walk();
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        int n[6], b[6], R = 0;
        unsigned s[6], nmi = 0;
        const int got = sscanf(line.c_str(), "%d %d %d %d %d %d %x %x %x %x %x %x %d %d %d %d %d %d %x %d", &n[0], &n[1], &n[2],
                               &n[3], &n[4], &n[5], &s[0], &s[1], &s[2], &s[3], &s[4], &s[5], &b[0], &b[1], &b[2], &b[3], &b[4],
                               &b[5], &nmi, &R);
        if (got != 20) {
            fprintf(stderr, "bad input line: %s\n", line.c_str());
            return 2;
        }
        NmiSearchKernel k(n[0], n[1], n[2], n[3], n[4], n[5], from_bits(s[0]), from_bits(s[1]), from_bits(s[2]), from_bits(s[3]),
                          from_bits(s[4]), from_bits(s[5]));
        k.setBest(b[0], b[1], b[2], b[3], b[4], b[5], from_bits(nmi));
        std::ostringstream before;
        before << k;
        printf("%d", k.isMiddle() ? 1 : 0);
        state(k);
        for (int r = 0; r < R; ++r) {
            k.resizeKernel();
            state(k);
        }
        std::ostringstream after;
        after << k;
        printf("\n%s\n%s\n", before.str().c_str(), after.str().c_str());
    }
    return 0;
}
