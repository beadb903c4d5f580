// Test harness (tests/test_host_units.py): cdf_search_rounds (vecmath.h: MCPT_CDF_LEVELS bisection levels per round trip, the form the
// device uses for an environment map's tables) returns cdf_search's index for ANY table — monotone, with runs of equal entries,
// not monotone at all (the reference's quirk Q7 searches such tables), with NaNs — and any target, table entries included.
#include "vecmath.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main()
{
    srand(7);
    long differ = 0, n = 0;
    for (int trial = 0; trial < 4000; ++trial)
    {
        const uint32_t num = 2 + rand() % 3000;
        std::vector<float> t(num);
        const int mode = trial % 4;
        float acc = 0;
        for (uint32_t i = 0; i < num; ++i)
        {
            if (mode == 0)
                acc += (rand() % 5 == 0) ? 0.0f : (rand() % 1000) / 1000.0f, t[i] = acc;
            else if (mode == 1)
                t[i] = (rand() % 64) / 64.0f;
            else if (mode == 2)
                t[i] = std::sin(i * 0.01f);
            else
                t[i] = (rand() % 100 == 0) ? NAN : static_cast<float>(i) / num;
        }
        if (mode == 0)
            for (auto &v : t)
                v /= (acc > 0 ? acc : 1);
        for (int q = 0; q < 400; ++q)
        {
            const float target = (q % 3 == 0) ? t[rand() % num] : (rand() % 4096) / 4096.0f;
            ++n;
            if (mcpt::cdf_search(num, t.data(), target) != mcpt::cdf_search_rounds(num, t.data(), target))
                ++differ;
        }
    }
    printf("%ld searches, %ld differ\n", n, differ);
    return differ == 0 ? 0 : 1;
}
