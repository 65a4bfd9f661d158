// nmi_search_kernel.hpp -- NmiSearchKernel with the reference's public interface
// (Thirdparty/Localization/nmiSearchKernel.hpp:25-86) on top of the C struct of include/nmi_host.h, so that
// code written against the reference class (Tracking.cc:1905-1956,2001-2130; localization.cpp:390-420)
// compiles unchanged while the logic lives in one place (nmi_driver.cpp).
#pragma once
#include <ostream>

#include "nmi_config.hpp"
#include "nmi_host.h"

class NmiSearchKernel {
public:
    // same public data members as the reference (nmiSearchKernel.hpp:29-35)
    int numSynthX, numSynthY, numSynthZ, numWarpX, numWarpY, numWarpZ;
    float stepX, stepY, stepZ, stepRadX, stepRadY, stepRadZ;
    float NMI;
    int bestSynthX, bestSynthY, bestSynthZ, bestWarpX, bestWarpY, bestWarpZ;

    NmiSearchKernel() { from_c(blank()); }
    NmiSearchKernel(int numsynthx, int numsynthy, int numsynthz, int numwarpx, int numwarpy, int numwarpz, float stepx,
                    float stepy, float stepz, float stepradx, float steprady, float stepradz)
    {
        from_c(blank());
        setKernel(numsynthx, numsynthy, numsynthz, numwarpx, numwarpy, numwarpz, stepx, stepy, stepz, stepradx, steprady,
                  stepradz);
    }
    explicit NmiSearchKernel(const nmi_search_kernel &c) { from_c(c); }

    void setKernel(int numsynthx, int numsynthy, int numsynthz, int numwarpx, int numwarpy, int numwarpz, float stepx,
                   float stepy, float stepz, float stepradx, float steprady, float stepradz)
    {
        numSynthX = numsynthx, numSynthY = numsynthy, numSynthZ = numsynthz;
        numWarpX = numwarpx, numWarpY = numwarpy, numWarpZ = numwarpz;
        stepX = stepx, stepY = stepy, stepZ = stepz;
        stepRadX = stepradx, stepRadY = steprady, stepRadZ = stepradz;
    }
    void setKernel(NmiSearchKernel *o)
    {
        setKernel(o->numSynthX, o->numSynthY, o->numSynthZ, o->numWarpX, o->numWarpY, o->numWarpZ, o->stepX, o->stepY,
                  o->stepZ, o->stepRadX, o->stepRadY, o->stepRadZ);
    }
    void setBest(int bestsynthx, int bestsynthy, int bestsynthz, int bestwarpx, int bestwarpy, int bestwarpz, float nmi)
    {
        bestSynthX = bestsynthx, bestSynthY = bestsynthy, bestSynthZ = bestsynthz;
        bestWarpX = bestwarpx, bestWarpY = bestwarpy, bestWarpZ = bestwarpz;
        NMI = nmi;
    }
    void setBest(NmiSearchKernel *o)  // copies the indices only, like the reference overload (nmiSearchKernel.cpp:83-91)
    {
        bestSynthX = o->bestSynthX, bestSynthY = o->bestSynthY, bestSynthZ = o->bestSynthZ;
        bestWarpX = o->bestWarpX, bestWarpY = o->bestWarpY, bestWarpZ = o->bestWarpZ;
    }
    void setTo(NmiSearchKernel *o)
    {
        setKernel(o);
        setBest(o);
        NMI = o->NMI;
    }
    bool isMiddle()
    {
        const nmi_search_kernel c = to_c();
        return nmi_sk_is_middle(&c) != 0;
    }
    void resizeKernel()
    {
        nmi_search_kernel c = to_c();
        nmi_properties p;
        nmi_properties_default(&p);
        p.step_factor = nmi_prop_STEPFACTOR;
        p.min_kernel_rotation = nmi_prop_MIN_KERNEL_ROTATION;
        p.min_kernel_translation = nmi_prop_MIN_KERNEL_TRANSLATION;
        nmi_sk_resize(&c, &p);
        from_c(c);
    }
    void resetKernel() { setKernel(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1); }
    void resetBest() { setBest(-1, -1, -1, -1, -1, -1, 0); }
    void reset() { from_c(blank()); }

    // the reference's 19 read accessors (nmiSearchKernel.hpp:66-84), one per public field
#define NMI_SK_GET(type, Name, field) \
    type get##Name() { return field; }
    NMI_SK_GET(int, NumSynthX, numSynthX)
    NMI_SK_GET(int, NumSynthY, numSynthY)
    NMI_SK_GET(int, NumSynthZ, numSynthZ)
    NMI_SK_GET(int, NumWarpX, numWarpX)
    NMI_SK_GET(int, NumWarpY, numWarpY)
    NMI_SK_GET(int, NumWarpZ, numWarpZ)
    NMI_SK_GET(float, StepX, stepX)
    NMI_SK_GET(float, StepY, stepY)
    NMI_SK_GET(float, StepZ, stepZ)
    NMI_SK_GET(float, StepRadX, stepRadX)
    NMI_SK_GET(float, StepRadY, stepRadY)
    NMI_SK_GET(float, StepRadZ, stepRadZ)
    NMI_SK_GET(int, BestSynthX, bestSynthX)
    NMI_SK_GET(int, BestSynthY, bestSynthY)
    NMI_SK_GET(int, BestSynthZ, bestSynthZ)
    NMI_SK_GET(int, BestWarpX, bestWarpX)
    NMI_SK_GET(int, BestWarpY, bestWarpY)
    NMI_SK_GET(int, BestWarpZ, bestWarpZ)
    NMI_SK_GET(float, Nmi, NMI)
#undef NMI_SK_GET

    nmi_search_kernel to_c() const
    {
        nmi_search_kernel c;
        const int n[6] = {numSynthX, numSynthY, numSynthZ, numWarpX, numWarpY, numWarpZ};
        const float s[6] = {stepX, stepY, stepZ, stepRadX, stepRadY, stepRadZ};
        const int b[6] = {bestSynthX, bestSynthY, bestSynthZ, bestWarpX, bestWarpY, bestWarpZ};
        for (int a = 0; a < 6; ++a) c.num[a] = n[a], c.step[a] = s[a], c.best[a] = b[a];
        c.nmi = NMI;
        return c;
    }
    void from_c(const nmi_search_kernel &c)
    {
        setKernel(c.num[0], c.num[1], c.num[2], c.num[3], c.num[4], c.num[5], c.step[0], c.step[1], c.step[2], c.step[3],
                  c.step[4], c.step[5]);
        setBest(c.best[0], c.best[1], c.best[2], c.best[3], c.best[4], c.best[5], c.nmi);
    }

private:
    static nmi_search_kernel blank()
    {
        nmi_search_kernel c;
        nmi_sk_init(&c);
        return c;
    }
};

inline std::ostream &operator<<(std::ostream &os, const NmiSearchKernel &k)
{
    char buf[512];
    const nmi_search_kernel c = k.to_c();
    nmi_sk_format(&c, buf, sizeof buf);
    return os << buf;
}
