// mcpt_cli: command-line driver over the C ABI of include/mcpt.h.
//
// Same switches as the reference program (apps/main.cpp:98-199):
//   [-g/--gpu] -i <scene> [-o <image>] [-w <width>] [-h <height>] [-s <spp>]
// <scene> is a Mitsuba-style .xml file, a .mcsd configuration, or
// "builtin:<name>".  Differences, all deliberate:
//   * -g/--gpu is the default; --gpus N cuts the frame over N GPUs of this node (8x8 tiles round-robin, one
//     RCCL gather, mcpt_tiled_renderer_*); -c/--cpu runs the kernel body on host threads through the optional
//     libmcpt_host.so (include/mcpt_host.h) found next to this program — libmcpt_hip.so itself has no CPU
//     path; -p/--preview (the GLUT viewer) is not part of this program;
//   * the output format follows the suffix (.png .exr .pfm .f32) instead of
//     being forced to .png; the default stays "result.png";
//   * the exit status is non-zero on failure (the reference always returns 0).
#include <chrono>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcpt.h"

namespace
{

void Usage()
{
    std::fprintf(stderr,
                 "mcpt_cli (%s)\n\n"
                 "  mcpt_cli [-g|--gpu] -i|--input <scene.xml | scene.mcsd | builtin:cornell-box>\n"
                 "           [-o|--output <result.png|.exr|.pfm|.f32>] [-w|--width N] [-h|--height N]\n"
                 "           [-s|--spp N] [-d|--device N] [--gpus N] [-c|--cpu] [--threads N]\n"
                 "           [--save-config <file.mcsd>] [--standins <table.txt>] [--rng reference|pcg|sobol] [--seed N]\n"
                 "           [--check-walks]\n\n"
                 "  --standins   procedural stand-ins for mesh files the scene names but that are not on disk\n"
                 "  --gpu        render with HIP on the selected device (the default)\n"
                 "  --gpus N     cut the frame over HIP devices 0 .. N-1 (one RCCL gather to device 0)\n"
                 "  --cpu        render on host threads (needs libmcpt_host.so next to this program)\n"
                 "  --rng        reference (default): the reference's random stream, one per pixel through all its samples\n"
                 "               (include/csrt/utils/math.hpp:43-63, renderer.cpp:62-81): frames comparable per pixel;\n"
                 "               pcg: throughput mode, an independent PCG-hashed stream per (pixel, sample) from --seed;\n"
                 "               sobol: the same with Owen-scrambled Sobol points (at most 8192 spp)\n"
                 "  --check-walks  (one GPU) render the whole film with the production ray query AND with the reference's visiting order\n"
                 "               first; pixels that differ are reported and the frame is rendered with the reference-order walk\n",
                 mcpt_version());
}

bool EndsWith(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int Fail(const char *what)
{
    std::fprintf(stderr, "[error] %s\n\t%s\n", what, mcpt_last_error());
    return 1;
}

} // namespace

int main(int argc, char **argv)
{
    std::string input, output = "result.png", save_config, standins_file;
    int width = 0, height = 0, spp = 0, device = 0, gpus = 1, threads = 0;
    bool on_cpu = false;
    int rng_mode = 0; // --rng reference | pcg | sobol (mcpt_renderer_set_rng)
    bool check_walks = false; // --check-walks: the user's whole film with the production ray query and with the reference-order walk
    unsigned rng_seed = 1;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        const bool has_value = i + 1 < argc;
        if (a == "--cpu" || a == "-c")
            on_cpu = true;
        else if (a == "--gpus" && has_value)
            gpus = std::atoi(argv[++i]);
        else if (a == "--threads" && has_value)
            threads = std::atoi(argv[++i]);
        else if (a == "--preview" || a == "-p")
        {
            std::fprintf(stderr, "[error] --preview: the interactive viewer is not part of this program.\n");
            return 2;
        }
        else if (a == "--gpu" || a == "-g")
            ;
        else if ((a == "--width" || a == "-w") && has_value)
            width = std::atoi(argv[++i]);
        else if ((a == "--height" || a == "-h") && has_value)
            height = std::atoi(argv[++i]);
        else if ((a == "--spp" || a == "-s") && has_value)
            spp = std::atoi(argv[++i]);
        else if ((a == "--device" || a == "-d") && has_value)
            device = std::atoi(argv[++i]);
        else if ((a == "--input" || a == "-i") && has_value)
            input = argv[++i];
        else if ((a == "--output" || a == "-o") && has_value)
            output = argv[++i];
        else if (a == "--save-config" && has_value)
            save_config = argv[++i];
        else if (a == "--standins" && has_value)
            standins_file = argv[++i];
        else if (a == "--rng" && has_value)
        {
            const std::string v = argv[++i];
            if (v == "reference")
                rng_mode = 0;
            else if (v == "pcg")
                rng_mode = 1;
            else if (v == "sobol")
                rng_mode = 2;
            else
            {
                std::fprintf(stderr, "[error] --rng: reference (the reference's per-pixel stream: frames comparable per pixel), pcg "
                                     "(throughput mode: an independent PCG-hashed stream per sample) or sobol (throughput mode with "
                                     "Owen-scrambled Sobol points, at most 8192 spp).\n");
                return 2;
            }
        }
        else if (a == "--check-walks")
            check_walks = true;
        else if (a == "--seed" && has_value)
            rng_seed = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        else if (a == "--help")
        {
            Usage();
            return 0;
        }
        else
        {
            std::fprintf(stderr, "[error] unknown or incomplete option '%s'.\n", a.c_str());
            Usage();
            return 2;
        }
    }
    if (input.empty())
    {
        Usage();
        return 2;
    }

    mcpt_config *config = nullptr;
    int rc;
    if (input.rfind("builtin:", 0) == 0)
        rc = mcpt_config_builtin(input.c_str() + 8, &config);
    else if (EndsWith(input, ".mcsd"))
        rc = mcpt_config_load_mcsd(input.c_str(), &config);
    else if (!standins_file.empty())
    {
        std::string table;
        if (FILE *f = std::fopen(standins_file.c_str(), "rb"))
        {
            char buf[4096];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof buf, f)) > 0)
                table.append(buf, n);
            std::fclose(f);
        }
        else
        {
            std::fprintf(stderr, "[error] cannot read '%s'.\n", standins_file.c_str());
            return 2;
        }
        rc = mcpt_config_load_xml_with_standins(input.c_str(), table.c_str(), &config);
    }
    else
        rc = mcpt_config_load_xml(input.c_str(), &config);
    if (rc != 0)
        return Fail("cannot load the scene.");
    if (mcpt_config_set_film(config, width, height, spp) != 0)
        return Fail("invalid film override.");
    if (!save_config.empty() && mcpt_config_save_mcsd(config, save_config.c_str()) != 0)
        return Fail("cannot save the configuration.");
    mcpt_config_get_film(config, &width, &height, &spp);

    std::vector<float> frame(static_cast<size_t>(width) * height * 3);
    const auto t0 = std::chrono::steady_clock::now();
    if (rng_mode != 0 && (on_cpu || gpus > 1))
    {
        std::fprintf(stderr, "[error] --rng pcg | sobol are modes of the single-GPU renderer (the host path and the tiled renderer keep the reference's stream).\n");
        return 2;
    }
    if (check_walks && (on_cpu || gpus > 1))
    {
        // (round 5's advisor: the flag used to be accepted and ignored here — a user would believe the scene was verified)
        std::fprintf(stderr, "[error] --check-walks compares the single-GPU renderer's ray query with the reference-order walk: not with --cpu (which IS the reference-order "
                             "walk) or --gpus N (check the scene once on one GPU).\n");
        return 2;
    }
    if (on_cpu)
    {
        // the reference's `--cpu` (apps/main.cpp:130-137): the kernel body on host threads, from the optional
        // host library next to this executable
        char self[4096];
        const ssize_t len = readlink("/proc/self/exe", self, sizeof self - 1);
        std::string dir = len > 0 ? std::string(self, static_cast<size_t>(len)) : std::string("./mcpt_cli");
        dir.erase(dir.find_last_of('/') == std::string::npos ? 0 : dir.find_last_of('/'));
        void *host = dlopen((dir + "/libmcpt_host.so").c_str(), RTLD_NOW);
        if (!host)
        {
            std::fprintf(stderr, "[error] --cpu: cannot load libmcpt_host.so (%s).\n", dlerror());
            return 1;
        }
        auto render = reinterpret_cast<int (*)(const void *, size_t, int, float *, double *)>(dlsym(host, "mcpt_host_render"));
        auto last_error = reinterpret_cast<const char *(*)()>(dlsym(host, "mcpt_host_last_error"));
        size_t size = 0;
        if (!render || !last_error || mcpt_config_serialize(config, nullptr, 0, &size) != 0)
            return Fail("--cpu: incomplete host library.");
        std::vector<unsigned char> bytes(size);
        if (mcpt_config_serialize(config, bytes.data(), bytes.size(), &size) != 0)
            return Fail("cannot serialise the configuration.");
        mcpt_config_destroy(config);
        double seconds = 0;
        if (render(bytes.data(), bytes.size(), threads, frame.data(), &seconds) != 0)
        {
            std::fprintf(stderr, "[error] %s\n", last_error());
            return 1;
        }
        std::fprintf(stderr, "[info] %d x %d, %d spp on host threads: draw %.3f s (%.2f Msamples/s)\n", width, height, spp,
                     seconds, seconds > 0 ? double(width) * height * spp / seconds / 1e6 : 0.0);
    }
    else if (gpus > 1)
    {
        mcpt_tiled_renderer *tiled = nullptr;
        std::vector<int> devices;
        for (int k = 0; k < gpus; ++k)
            devices.push_back(device + k);
        rc = mcpt_tiled_renderer_create(config, gpus, devices.data(), 0, &tiled);
        mcpt_config_destroy(config);
        if (rc != 0)
            return Fail("error when create renderer.");
        const auto t1 = std::chrono::steady_clock::now();
        mcpt_stats stats;
        rc = mcpt_tiled_renderer_draw(tiled, frame.data(), &stats);
        mcpt_tiled_renderer_destroy(tiled);
        if (rc != 0)
            return Fail("error when draw.");
        std::fprintf(stderr, "[info] %d x %d, %d spp on %d GPUs: scene commit %.3f s, draw %.3f s (%.1f Msamples/s)\n", width,
                     height, spp, gpus, std::chrono::duration<double>(t1 - t0).count(), stats.render_seconds,
                     stats.render_seconds > 0 ? double(width) * height * spp / stats.render_seconds / 1e6 : 0.0);
    }
    else
    {
        mcpt_renderer *renderer = nullptr;
        rc = mcpt_renderer_create(config, device, &renderer);
        mcpt_config_destroy(config);
        if (rc != 0)
            return Fail("error when create renderer.");
        if (rng_mode != 0 && mcpt_renderer_set_rng(renderer, rng_mode, rng_seed, 0) != 0)
        {
            mcpt_renderer_destroy(renderer);
            return Fail("--rng: cannot select the random-number mode.");
        }
        if (check_walks)
        {
            // The production ray query (ordered / pool walk: its tie radius and sliver reach are engineering bounds)
            // against the reference's own visiting order on THIS scene, every pixel, full spp: a scene outside the bounds is reported
            // and rendered with the reference-order walk (slower, the reference's image).  mcpt_renderer_create checks a sample of
            // the film at 1 spp by default; this is the thorough form.
            uint64_t differing = 0;
            uint32_t first = 0;
            float worst = 0.0f;
            if (mcpt_renderer_check_walks(renderer, &differing, &first, &worst) != 0)
            {
                mcpt_renderer_destroy(renderer);
                return Fail("--check-walks: the comparison could not run.");
            }
            if (differing != 0)
            {
                std::fprintf(stderr, "[warning] --check-walks: %llu pixel(s) differ between the production walk and the reference-order walk "
                                     "(first: pixel %u, largest difference %g): rendering with the reference-order walk.\n",
                             static_cast<unsigned long long>(differing), first, worst);
                mcpt_renderer_set_walk(renderer, 1);
            }
            else
                std::fprintf(stderr, "[info] --check-walks: both walks agree on every pixel.\n");
        }
        const auto t1 = std::chrono::steady_clock::now();
        mcpt_stats stats;
        if (mcpt_renderer_draw(renderer, frame.data(), &stats) != 0)
        {
            mcpt_renderer_destroy(renderer);
            return Fail("error when draw.");
        }
        mcpt_renderer_destroy(renderer);
        const double setup_s = std::chrono::duration<double>(t1 - t0).count();
        std::fprintf(stderr, "[info] %d x %d, %d spp: scene commit %.3f s, draw %.3f s (%.1f Msamples/s)\n", width, height,
                     spp, setup_s, stats.kernel_milliseconds * 1e-3,
                     stats.kernel_milliseconds > 0 ? double(width) * height * spp / (stats.kernel_milliseconds * 1e3) : 0.0);
    }
    if (mcpt_write_image(output.c_str(), frame.data(), width, height) != 0)
        return Fail("cannot write the image.");
    std::fprintf(stderr, "[info] save result as image \"%s\".\n", output.c_str());
    return 0;
}
