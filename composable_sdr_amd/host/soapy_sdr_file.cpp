// soapy-sdr's file-input mode (apps/SoapySDR.hs:181-283) on the C-ABI chain:
//   soapy_sdr_file --filename in.cf32 -n N -c M [--demod DeNo|DeNBFM kf|DeWBFM decim|DeAM] [-a dB] [-m] [-o output] [--chunksize 1024]
//                  [-s samplerate] [-b bandwidth] [--offset Hz] [--audio AU|WAV]
//                  [--world W --rank R --id-file PATH [--id-nonce N] [--device D]]
// --world W: one process per GPU, each reading the same file; process R owns the channels R, R + W, ... (interleaved channel shard,
// SURVEY 8e(A)) and writes only their <out>_ch<k+1> files (SoapySDR.hs:209-212); with --mix the partial sums of the W processes meet
// in one RCCL all-reduce per chunk (csdr_chain_process_mix) and process 0 writes the one mixed file (SoapySDR.hs:217-222).
// readFromFile -> [mixDown/mixUp (--offset)] -> [resampler (-b)] -> takeNArr -> compact(4*M*1024) -> fused chain (dcBlocker + PFB + demod [+mix]) -> fileSinks
// named <out>.cf32 / <out>_ch<k>.cf32 (DeNo, SoapySDR.hs:240) or raw .f32 for FM (the reference wraps
// the same samples in WAV/AU through libsndfile).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <cstring>
#include <iostream>

#include "csdr_host.hpp"

using namespace csdrhost;

struct FrontOpts { double samplerate = 2.56e6, bandwidth = 0.0, offset = 0.0; std::string audio; };
static FrontOpts g_front;

template <class Out> static int run(const std::string &in, const ChainOpts &o, size_t n, const std::string &out, size_t chunk, const char *ext)
{
    const uint32_t M = o.channels;
    const bool mixed = o.mix && M > 1;
    std::vector<std::shared_ptr<Fold<Array<Out>>>> sinks;
    auto make = [&](const std::string &stem) -> std::shared_ptr<Fold<Array<Out>>> {
        if constexpr (std::is_same<Out, float>::value) {
            if (!g_front.audio.empty()) {
                // getAudioSink decim fmt 1 (SoapySDR.hs:232-234): rate = round outBW `div` decim `div` nch
                const double bw = g_front.bandwidth != 0.0 ? g_front.bandwidth : g_front.samplerate;
                const uint32_t sr = (uint32_t)std::llround(bw) / (o.wbfm ? o.decim : 1u) / M;
                return std::make_shared<AudioFileSink>(g_front.audio, sr, 1u, stem);
            }
        }
        return std::make_shared<FileSink<Out>>(stem + ext);
    };
    struct NullSink : Fold<Array<Out>> { void step(const Array<Out> &) override {} void done() override {} };
    if (mixed || M == 1) {
        // the reference has ONE sink behind `mix`: of a sharded run, rank 0 writes it (every rank holds the same sum)
        if (o.rank == 0) sinks.push_back(make(out));
        else sinks.push_back(std::make_shared<NullSink>());
    }
    else for (uint32_t row = 0; row < o.owned(); row++) sinks.push_back(make(out + "_ch" + std::to_string(o.channel_of(row) + 1)));
    auto fold = compact<cf32>((size_t)4 * M * 1024, addPipe(fusedChain<Out>(o), std::static_pointer_cast<Fold<std::vector<Array<Out>>>>(
                                                                                    std::make_shared<Distribute<Out>>(sinks))));
    FILE *f = std::fopen(in.c_str(), "rb");
    if (!f) { std::cerr << "Unable to open source: " << in << "\n"; return 1; }
    // prep = takeNArr ns . (resampler . offset)  (SoapySDR.hs:190-207)
    using CPipe = Pipe<Array<cf32>, Array<cf32>>;
    const float fo = (float)(2.0 * 3.14159265358979323846 * g_front.offset / g_front.samplerate);
    CPipe offset = fo > 0 ? mixDown(fo, (uint32_t)chunk) : (fo < 0 ? mixUp(-fo, (uint32_t)chunk) : idPipe<Array<cf32>>());
    CPipe resamp = g_front.bandwidth != 0.0 ? resampler((float)(g_front.bandwidth / g_front.samplerate), 60.0f, (uint32_t)chunk)
                                            : idPipe<Array<cf32>>();
    auto prep = unPipe(compose(resamp, offset));            // (process, cleanup) <- unPipe (resampler . offset)
    TakeN take(n);
    Array<cf32> a(chunk);
    while (true) {
        a.resize(chunk);
        const size_t got = std::fread(a.data(), sizeof(cf32), chunk, f);
        if (!got) break;
        a.resize(got);
        Array<cf32> b = prep.process(a);
        if (!take.feed(b)) break;
        fold->step(b);
    }
    std::fclose(f);
    fold->done();
    prep.cleanup();
    return 0;
}

int main(int argc, char **argv)
{
    std::string in, out = "output", demod = "DeNo";
    ChainOpts o; o.flags = 0;
    size_t n = 1024, chunk = 1024;
    std::string id_file; int device = -1; uint64_t id_nonce = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; std::exit(2); } return argv[++i]; };
        if (a == "--filename") in = next();
        else if (a == "-n" || a == "--numsamples") n = std::strtoull(next(), nullptr, 10);
        else if (a == "-c" || a == "--channels") o.channels = (uint32_t)std::atoi(next());
        else if (a == "-a" || a == "--agc") o.agc = (float)std::atof(next());
        else if (a == "-m" || a == "--mix") o.mix = true;
        else if (a == "-o" || a == "--output") out = next();
        else if (a == "--chunksize") chunk = std::strtoull(next(), nullptr, 10);
        else if (a == "-s" || a == "--samplerate") g_front.samplerate = std::atof(next());
        else if (a == "-b" || a == "--bandwidth") g_front.bandwidth = std::atof(next());
        else if (a == "--offset") g_front.offset = std::atof(next());
        else if (a == "--audio") g_front.audio = next();
        else if (a == "--world") o.world = (uint32_t)std::atoi(next());
        else if (a == "--rank") o.rank = (uint32_t)std::atoi(next());
        else if (a == "--id-file") id_file = next();
        else if (a == "--id-nonce") id_nonce = std::strtoull(next(), nullptr, 10);
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--demod") { demod = next(); if (demod == "DeNBFM") { o.fm = true; o.kf = (float)std::atof(next()); } else if (demod == "DeAM") o.am = true; else if (demod == "DeWBFM") { o.wbfm = true; o.decim = (uint32_t)std::atoi(next()); } }
        else { std::cerr << "unknown option " << a << "\n"; return 2; }
    }
    if (in.empty()) { std::cerr << "--filename is required (SoapySDR live sources are out of scope)\n"; return 2; }
    if (o.world < 1 || o.rank >= o.world) { std::cerr << "--rank must be below --world\n"; return 2; }
    if (o.world > 1 && o.channels % o.world) { std::cerr << "--world must divide -c\n"; return 2; }
    if (o.world > 1 && o.mix && id_file.empty()) { std::cerr << "--world with --mix needs --id-file (the communicator's bootstrap)\n"; return 2; }
    try {
        o.device = device;
        // the communicator exists only where the path has an exchange step: --mix over channel shards (also a world of one, which
        // then runs the same C entry points)
        if (o.mix && o.channels > 1 && !id_file.empty()) o.comm = commFromIdFile(id_file, (int)o.rank, (int)o.world, device, id_nonce);
        if (o.wbfm) o.deemph_fc = (float)(5000.0 / (g_front.bandwidth != 0.0 ? g_front.bandwidth : g_front.samplerate));
        const int rc = (o.fm || o.am || o.wbfm) ? run<float>(in, o, n, out, chunk, ".f32") : run<cf32>(in, o, n, out, chunk, ".cf32");
        if (o.comm) check(csdr_comm_destroy(o.comm));
        return rc;
    } catch (const std::exception &e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
}
