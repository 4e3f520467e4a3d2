// dpt_amd: the reference's process surface for the LMC path (`dpt [--seedoffset N] scene.xml ...`,
// /root/reference/src/main.cpp:30-118) on top of the C ABI of liblmc_hip.so.  Same scene XML, same <dpt> keys, same
// output naming (<film filename>_timeuse_<seconds>s.exr written next to the scene, mlt.cpp:208) and the same stdout
// lines ("Average brightness:", "Elapsed time:", "Done!").  Only integrator = mcmc with mala = true (the LMC path) is
// served; anything else is refused.  Extra flags (not in the reference): --chains N (Markov chains resident on the GPUs,
// default: <dpt numchains>), --init-threads V (MLTInit streams, default 65536), --device D, --force-diffuse, --maxdepth D, and
// --gpus N (devices 0 .. N-1) / --devices a,b,.. (an explicit list; a device may appear more than once: bring-up on one GPU): the chains are
// sharded over the listed devices as ranks of ONE job (lmc_group_*: MLTInit sharded by init stream, contiguous chain-id ranges, the gradient
// cache's pushes exchanged while it fills -- the trajectories of a single device holding all the chains), and the per-device films are summed
// on the devices before the image is written (mlt.cpp:203-207 merges its per-thread films the same way).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lmc_abi.h"

static double Opt(lmc_ctx *ctx, const char *name) {
    double v = 0;
    if (lmc_get_option(ctx, name, &v) != 0) {
        fprintf(stderr, "%s\n", lmc_last_error());
        exit(1);
    }
    return v;
}

int main(int argc, char *argv[]) {
    if (argc <= 1) return 0;
    printf("Langevin MCMC dpt (MI355X back end)\n");
    int seedoffset = 0, device = 0, forceDiffuse = 0, maxDepth = 0, initThreads = 65536, maxDervDepth = 8;
    long long chains = 0;
    std::vector<int> devices;
    std::vector<std::string> filenames;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--seedoffset") seedoffset = std::stoi(argv[++i]);
        else if (a == "--max-derivatives-depth") maxDervDepth = std::stoi(argv[++i]);  // main.cpp:59-60: techniques longer than this get isotropic proposals
        else if (a == "--compile-pathlib" || a == "--compile-bidirpathlib" || a == "--compile-bidirpathlib2") {
            printf("%s: nothing to compile, the path programs are part of liblmc_hip.so\n", a.c_str());
        } else if (a == "--chains") chains = std::stoll(argv[++i]);
        else if (a == "--init-threads") initThreads = std::stoi(argv[++i]);
        else if (a == "--device") device = std::stoi(argv[++i]);
        else if (a == "--gpus") {
            const int n = std::stoi(argv[++i]);
            devices.clear();
            for (int d = 0; d < n; d++) devices.push_back(d);
        } else if (a == "--devices") {
            devices.clear();
            std::string list = argv[++i];
            for (size_t b = 0; b <= list.size();) {
                const size_t e = list.find(',', b) == std::string::npos ? list.size() : list.find(',', b);
                if (e > b) devices.push_back(std::stoi(list.substr(b, e - b)));
                b = e + 1;
            }
        } else if (a == "--force-diffuse") forceDiffuse = 1;
        else if (a == "--maxdepth") maxDepth = std::stoi(argv[++i]);
        else filenames.push_back(a);
    }
    if (devices.empty()) devices.push_back(device);
    const int visible = lmc_device_count();
    for (int d : devices)
        if (d < 0 || d >= visible) {
            fprintf(stderr, "device %d requested, %d HIP device(s) visible\n", d, visible);
            return 2;
        }
    const int nDev = (int)devices.size();
    for (const std::string &filename : filenames) {
        std::vector<lmc_ctx *> ctxs;
        for (int d : devices) {
            lmc_scene_desc desc;
            memset(&desc, 0, sizeof(desc));
            desc.scene_xml = filename.c_str();
            desc.force_diffuse = forceDiffuse, desc.max_depth = maxDepth, desc.seed_offset = seedoffset, desc.device = d, desc.use_gradient = 1;
            lmc_ctx *c = lmc_create(&desc);
            if (!c) {
                fprintf(stderr, "%s\n", lmc_last_error());
                return 1;
            }
            lmc_set_option(c, "max-derivatives-depth", maxDervDepth);
            ctxs.push_back(c);
        }
        lmc_ctx *ctx = ctxs[0];
        if (Opt(ctx, "mala") == 0 && Opt(ctx, "h2mc") == 0) {
            fprintf(stderr, "dpt_amd serves the LMC path only (<dpt> integrator=mcmc with mala=true or h2mc=true)\n");
            return 1;
        }
        int info[8];
        lmc_info(ctx, info);
        const int W = info[0], H = info[1];
        const int spp = (int)Opt(ctx, "spp"), directSpp = (int)Opt(ctx, "directspp");
        const long long numChains = chains > 0 ? chains : (long long)Opt(ctx, "numchains");
        // mlt.cpp:33-47
        printf("Compute direct lighting\n");
        if (lmc_direct_lighting(ctx, directSpp) != 0) {
            fprintf(stderr, "%s\n", lmc_last_error());
            return 1;
        }
        const long long totalSamples = (long long)spp * W * H;
        const long long numSamplesPerChain = totalSamples / numChains;
        const long long chainsNeedExtraSamples = numSamplesPerChain % numChains;  // (sic) mlt.cpp:40
        long long numInit = (long long)Opt(ctx, "numinitsamples");
        if (chains > 0 && numInit < 8 * numChains) {  // MLTInit needs at least as many contributions as chains (mlt.h:101-105)
            numInit = 8 * numChains;
            printf("numinitsamples raised to %lld for %lld chains\n", numInit, numChains);
        }
        if (nDev > 1) printf("%lld chains sharded over %d devices\n", numChains, nDev);
        if ((nDev == 1 ? lmc_chains_init(ctx, numInit, (int)numChains, initThreads, 0, (int)numChains, numSamplesPerChain, chainsNeedExtraSamples)
                       : lmc_group_chains_init(ctxs.data(), nDev, numInit, (int)numChains, initThreads, numSamplesPerChain, chainsNeedExtraSamples)) != 0) {
            fprintf(stderr, "%s\n", lmc_last_error());
            return 1;
        }
        float normalization = 0;
        long long nContribs = 0;
        lmc_init_result(ctx, &normalization, &nContribs);
        printf("Average brightness:%g\n", normalization);
        auto t0 = std::chrono::steady_clock::now();
        for (long long done = 0; done < numSamplesPerChain + 1; done += 64)
            if ((nDev == 1 ? lmc_chains_step(ctx, 64) : lmc_group_chains_step(ctxs.data(), nDev, 64)) != 0) {
                fprintf(stderr, "%s\n", lmc_last_error());
                return 1;
            }
        for (lmc_ctx *c : ctxs) lmc_sync(c);
        if (nDev > 1 && lmc_group_film_reduce(ctxs.data(), nDev, nullptr) != 0) {  // the per-device films summed on the devices (peer copies); timed with the loop
            fprintf(stderr, "%s\n", lmc_last_error());
            return 1;
        }
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Elapsed time:%g\n", elapsed);
        // MergeBuffer + BufferToFilm, mlt.cpp:203-207
        std::vector<float> direct((size_t)W * H * 3), indirect((size_t)W * H * 3), img((size_t)W * H * 3);
        lmc_direct_read(ctx, direct.data());
        lmc_film_read(ctx, indirect.data());
        const float dw = directSpp > 0 ? 1.0f / float(directSpp) : 0.0f, iw = spp > 0 ? 1.0f / float(spp) : 0.0f;
        for (size_t i = 0; i < img.size(); i++) img[i] = dw * direct[i] + iw * indirect[i];
        std::string dir = filename.rfind('/') != std::string::npos ? filename.substr(0, filename.rfind('/') + 1) : "";
        std::string out = dir + lmc_output_name(ctx) + "_timeuse_" + std::to_string(elapsed) + "s.exr";
        if (lmc_image_write_exr(out.c_str(), img.data(), W, H) != 0) {
            fprintf(stderr, "%s\n", lmc_last_error());
            return 1;
        }
        long long mutations = 0;
        for (lmc_ctx *c : ctxs) {
            long long st[8];
            double ws = 0;
            lmc_stats(c, st, &ws);
            mutations += st[0];
        }
        printf("%lld mutations, %.1f M mutations/s, wrote %s\n", mutations, mutations / elapsed * 1e-6, out.c_str());
        for (lmc_ctx *c : ctxs) lmc_destroy(c);
        printf("Done!\n");
    }
    return 0;
}
