// TEST INFRASTRUCTURE — wave_emu.cpp as a program (for the AddressSanitizer + UBSan build: tests/test_wave_emu.py):
//   wave_emu_asan <scene.mcsd> <features> <lds 0|1> <order> <poison 0|1> <poison word> <lane spread> <compact> <frame.f32> [<lds shortfall>]
// renders the scene with the kernel body of that instantiation in lockstep and writes the frame as raw float32.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Options
{
    uint32_t order, seed, poison, poison_word, max_blocks, per_cu, lane_spread, compact, scatter, threads, xcd_bands, lds_shortfall;
};
struct Report
{
    uint64_t collectives, rounds, queries;
    uint32_t blocks, lds_bytes, lane_spread, scatter;
};
extern "C" int mcpt_wave_emu_render(const char *, uint32_t, int, const Options *, float *, unsigned long long *, Report *);
extern "C" int mcpt_wave_emu_film(const char *, uint32_t *, uint32_t *);
extern "C" const char *mcpt_wave_emu_last_error(void);

int main(int argc, char **argv)
{
    if (argc != 10 && argc != 11)
    {
        fprintf(stderr, "usage: %s scene.mcsd features lds order poison poison_word lane_spread compact frame.f32\n", argv[0]);
        return 2;
    }
    uint32_t width = 0, height = 0;
    if (mcpt_wave_emu_film(argv[1], &width, &height) != 0)
    {
        fprintf(stderr, "%s\n", mcpt_wave_emu_last_error());
        return 1;
    }
    const Options opt{static_cast<uint32_t>(strtoul(argv[4], nullptr, 0)), 1u, static_cast<uint32_t>(strtoul(argv[5], nullptr, 0)), static_cast<uint32_t>(strtoul(argv[6], nullptr, 0)), 2u, 1u,
                      static_cast<uint32_t>(strtoul(argv[7], nullptr, 0)), static_cast<uint32_t>(strtoul(argv[8], nullptr, 0)), 0xFFFFFFFFu, 2u, 0u, argc == 11 ? static_cast<uint32_t>(strtoul(argv[10], nullptr, 0)) : 0u};
    std::vector<float> frame(size_t(width) * height * 3);
    Report rep{};
    if (mcpt_wave_emu_render(argv[1], static_cast<uint32_t>(strtoul(argv[2], nullptr, 0)), atoi(argv[3]), &opt, frame.data(), nullptr, &rep) != 0)
    {
        fprintf(stderr, "%s\n", mcpt_wave_emu_last_error());
        return 1;
    }
    FILE *f = fopen(argv[9], "wb");
    if (!f || fwrite(frame.data(), sizeof(float), frame.size(), f) != frame.size())
        return 1;
    fclose(f);
    printf("{\"width\": %u, \"height\": %u, \"blocks\": %u, \"lds_bytes\": %u, \"collectives\": %llu, \"queries\": %llu}\n", width, height, rep.blocks, rep.lds_bytes,
           static_cast<unsigned long long>(rep.collectives), static_cast<unsigned long long>(rep.queries));
    return 0;
}
