// C++ host side above the C ABI: the reference's block protocol and stream combinators
// (GHC is not available in this image, and the reference is compiled code, so the host-side
// mirror that ships is C++; composable_sdr_amd/*.py is the same thing for the tests).
//
//   Pipe<A,B>      Types.hs:51-55     { start, process, done } around one native handle
//   compose        Types.hs:93-99     process2 >=> process1, done2 before done1
//   idPipe         Types.hs:101-102   Category id
//   unPipe         Types.hs:109-115   create now; (per-array map, cleanup) for the stream side
//   Fold<A>        streamly Fold      step / done
//   addPipe        Types.hs:117-131   downstream start first, then create; done: destroy, then downstream
//   takeNArr       Trans.hs:33-56     pass arrays until n samples, trimming the last one
//   compact        Trans.hs:58-84     emit exactly n, keep the rest, flush the remainder at done
//   distribute_    Trans.hs:106-117   element k of the list to fold k (zip semantics)
//   mix            Trans.hs:119-122   left fold of element-wise +
//   fileSink       Sink.hs:29-34      raw chunk writer
//   readFromFile   Source.chs:259-271 raw CF32 chunks of <= n samples
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>

#include "../../include/csdr.h"

namespace csdrhost {

using cf32 = std::complex<float>;
template <class T> using Array = std::vector<T>;

struct CsdrError : std::runtime_error {
    int code;
    CsdrError(int c, const std::string &m) : std::runtime_error("csdr error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int code) { if (code != 0) throw CsdrError(code, csdr_last_error()); }   // Common.hs:32-33 `try`

// ---- Pipe -------------------------------------------------------------------------------
template <class A, class B> struct Pipe {
    std::function<std::shared_ptr<void>()> start;
    std::function<B(void *, const A &)> process;
    std::function<void(void *)> done;
};

template <class A, class B, class C> Pipe<A, C> compose(Pipe<B, C> p1, Pipe<A, B> p2)
{
    struct R { std::shared_ptr<void> r1, r2; };
    Pipe<A, C> p;
    p.start = [=]() { auto r = std::make_shared<R>(); r->r1 = p1.start(); r->r2 = p2.start(); return std::static_pointer_cast<void>(r); };
    p.process = [=](void *r, const A &a) { auto *rr = static_cast<R *>(r); return p1.process(rr->r1.get(), p2.process(rr->r2.get(), a)); };
    p.done = [=](void *r) { auto *rr = static_cast<R *>(r); p2.done(rr->r2.get()); p1.done(rr->r1.get()); };
    return p;
}

template <class A> Pipe<A, A> idPipe()
{
    Pipe<A, A> p;
    p.start = []() { return std::shared_ptr<void>(); };
    p.process = [](void *, const A &a) { return a; };
    p.done = [](void *) {};
    return p;
}

// unPipe (Types.hs:109-115): `r <- creat; return (S.mapM (process r), dest r)`.  The stream side here is a push
// loop, so the transformer is the per-array function S.mapM would apply; `cleanup` is run once the stream has been
// folded (SoapySDR.hs:206, :282).
template <class A, class B> struct UnPiped {
    std::function<B(const A &)> process;
    std::function<void()> cleanup;
};
template <class A, class B> UnPiped<A, B> unPipe(Pipe<A, B> p)
{
    std::shared_ptr<void> r = p.start();
    UnPiped<A, B> u;
    u.process = [p, r](const A &a) { return p.process(r.get(), a); };
    u.cleanup = [p, r]() { p.done(r.get()); };
    return u;
}

// ---- Fold -------------------------------------------------------------------------------
template <class A> struct Fold {
    virtual ~Fold() = default;
    virtual void step(const A &a) = 0;
    virtual void done() = 0;
};

template <class A, class B> struct AddPipe : Fold<A> {
    Pipe<A, B> pipe; std::shared_ptr<Fold<B>> down; std::shared_ptr<void> r;
    AddPipe(Pipe<A, B> p, std::shared_ptr<Fold<B>> d) : pipe(std::move(p)), down(std::move(d)), r(pipe.start()) {}
    void step(const A &a) override { down->step(pipe.process(r.get(), a)); }
    void done() override { pipe.done(r.get()); down->done(); }
};
template <class A, class B> std::shared_ptr<Fold<A>> addPipe(Pipe<A, B> p, std::shared_ptr<Fold<B>> d)
{
    return std::make_shared<AddPipe<A, B>>(std::move(p), std::move(d));
}

template <class T> struct Compact : Fold<Array<T>> {
    size_t n; std::shared_ptr<Fold<Array<T>>> down; Array<T> buf;
    Compact(size_t n_, std::shared_ptr<Fold<Array<T>>> d) : n(n_), down(std::move(d)) {}
    void step(const Array<T> &a) override {
        buf.insert(buf.end(), a.begin(), a.end());
        if (buf.size() >= n) {                              // emit EXACTLY n, keep the rest (even if >= n)
            Array<T> head(buf.begin(), buf.begin() + n);
            buf.erase(buf.begin(), buf.begin() + n);
            down->step(head);
        }
    }
    void done() override { down->step(buf); buf.clear(); down->done(); }   // remainder pushed even when empty
};
template <class T> std::shared_ptr<Fold<Array<T>>> compact(size_t n, std::shared_ptr<Fold<Array<T>>> d)
{
    return std::make_shared<Compact<T>>(n, std::move(d));
}

template <class T> struct Distribute : Fold<std::vector<Array<T>>> {
    std::vector<std::shared_ptr<Fold<Array<T>>>> folds;
    explicit Distribute(std::vector<std::shared_ptr<Fold<Array<T>>>> f) : folds(std::move(f)) {}
    void step(const std::vector<Array<T>> &as) override { for (size_t k = 0; k < folds.size() && k < as.size(); k++) folds[k]->step(as[k]); }
    void done() override { for (auto &f : folds) f->done(); }
};

template <class T> Array<T> mix(const std::vector<Array<T>> &chans)
{
    Array<T> acc = chans.at(0);
    for (size_t k = 1; k < chans.size(); k++) {
        if (chans[k].size() < acc.size()) acc.resize(chans[k].size());       // zipWith truncates
        for (size_t i = 0; i < acc.size(); i++) acc[i] = acc[i] + chans[k][i];
    }
    return acc;
}

// takeNArr as a push-side limiter: returns false once the stream is finished
struct TakeN {
    size_t n, seen = 0;
    explicit TakeN(size_t n_) : n(n_) {}
    template <class T> bool feed(Array<T> &a) {
        if (seen == n) return false;
        if (n - seen < a.size()) a.resize(n - seen);
        seen += a.size();
        return true;
    }
};

template <class T> struct FileSink : Fold<Array<T>> {
    FILE *f;
    explicit FileSink(const std::string &path) : f(std::fopen(path.c_str(), "wb")) { if (!f) throw std::runtime_error("cannot open " + path); }
    ~FileSink() override { if (f) std::fclose(f); }
    void step(const Array<T> &a) override { if (!a.empty() && std::fwrite(a.data(), sizeof(T), a.size(), f) != a.size()) throw std::runtime_error("short write"); }
    void done() override { if (f) { std::fclose(f); f = nullptr; } }
};

// audioFileSink fmt sr sn nch fp (Sink.hs:41-74): libsndfile float, big-endian, fp + ".au" / ".wav".  AU: 24-byte
// header (".snd", 24, data bytes, 6 = IEEE float, rate, channels).  WAV: libsndfile writes RIFX with a time-stamped
// PEAK chunk (not reproducible); this writer emits RIFX + fmt(tag 3) + fact + data.  Unverified against libsndfile.
struct AudioFileSink : Fold<Array<float>> {
    FILE *f = nullptr; bool au; uint32_t sr, nch; uint64_t nbytes = 0; std::string path;
    static void be32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
    static void be16(unsigned char *p, uint16_t v) { p[0] = v >> 8; p[1] = v & 255; }
    AudioFileSink(const std::string &fmt, uint32_t sr_, uint32_t nch_, const std::string &fp)
        : au(fmt == "AU" || fmt == "au"), sr(sr_), nch(nch_), path(fp + ((fmt == "AU" || fmt == "au") ? ".au" : ".wav"))
    {
        f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + path);
        header(au ? 0xffffffffu : 0u);
    }
    ~AudioFileSink() override { if (f) std::fclose(f); }
    void header(uint32_t n)
    {
        unsigned char h[56];
        std::fseek(f, 0, SEEK_SET);
        if (au) {
            std::memcpy(h, ".snd", 4); be32(h + 4, 24); be32(h + 8, n); be32(h + 12, 6); be32(h + 16, sr); be32(h + 20, nch);
            std::fwrite(h, 1, 24, f);
        } else {
            std::memcpy(h, "RIFX", 4); be32(h + 4, 4 + 24 + 12 + 8 + n); std::memcpy(h + 8, "WAVE", 4);
            std::memcpy(h + 12, "fmt ", 4); be32(h + 16, 16); be16(h + 20, 3); be16(h + 22, (uint16_t)nch); be32(h + 24, sr);
            be32(h + 28, sr * 4 * nch); be16(h + 32, (uint16_t)(4 * nch)); be16(h + 34, 32);
            std::memcpy(h + 36, "fact", 4); be32(h + 40, 4); be32(h + 44, n / (4 * nch));
            std::memcpy(h + 48, "data", 4); be32(h + 52, n);
            std::fwrite(h, 1, 56, f);
        }
    }
    void step(const Array<float> &a) override
    {
        std::vector<unsigned char> b(a.size() * 4);
        for (size_t i = 0; i < a.size(); i++) { uint32_t u; std::memcpy(&u, &a[i], 4); be32(b.data() + 4 * i, u); }
        if (!b.empty() && std::fwrite(b.data(), 1, b.size(), f) != b.size()) throw std::runtime_error("short write");
        nbytes += b.size();
    }
    void done() override { if (f) { header((uint32_t)nbytes); std::fclose(f); f = nullptr; } }
};

// ---- front-end Pipes: resampler r as (Liquid.chs:115-117), mixDown / mixUp f (Liquid.chs:805-809) ----
inline Pipe<Array<cf32>, Array<cf32>> resampler(float r, float as_db, uint32_t max_in)
{
    Pipe<Array<cf32>, Array<cf32>> p;
    p.start = [=]() {
        csdr_resamp *h = nullptr;
        check(csdr_resamp_create(r, as_db, max_in, &h));
        return std::shared_ptr<void>(h, [](void *q) { csdr_resamp_destroy(static_cast<csdr_resamp *>(q)); });
    };
    p.process = [](void *rr, const Array<cf32> &a) {
        auto *h = static_cast<csdr_resamp *>(rr);
        Array<cf32> y(csdr_resamp_max_out(h, (uint32_t)a.size()));          // 2*ceil(r*nx), Liquid.chs:81
        uint32_t n = 0;
        check(csdr_resamp_process(h, reinterpret_cast<const float *>(a.data()), (uint32_t)a.size(), reinterpret_cast<float *>(y.data()), &n));
        y.resize(n);                                                         // shrinkToFit, Liquid.chs:98
        return y;
    };
    p.done = [](void *) {};
    return p;
}
inline Pipe<Array<cf32>, Array<cf32>> ncoMixer(float f, bool up, uint32_t max_in)
{
    Pipe<Array<cf32>, Array<cf32>> p;
    p.start = [=]() {
        csdr_nco *h = nullptr;
        check(csdr_nco_create(f, max_in, &h));
        return std::shared_ptr<void>(h, [](void *q) { csdr_nco_destroy(static_cast<csdr_nco *>(q)); });
    };
    p.process = [up](void *rr, const Array<cf32> &a) {
        auto *h = static_cast<csdr_nco *>(rr);
        Array<cf32> y(a.size());
        if (!a.empty())
            check((up ? csdr_nco_mix_up : csdr_nco_mix_down)(h, reinterpret_cast<const float *>(a.data()), (uint32_t)a.size(), reinterpret_cast<float *>(y.data())));
        return y;
    };
    p.done = [](void *) {};
    return p;
}
inline Pipe<Array<cf32>, Array<cf32>> mixDown(float f, uint32_t max_in) { return ncoMixer(f, false, max_in); }
inline Pipe<Array<cf32>, Array<cf32>> mixUp(float f, uint32_t max_in) { return ncoMixer(f, true, max_in); }

// ---- the fused chain as a Pipe (replaces mix . mux (replicate nch demod) . firpfbchChannelizer nc) ----
struct ChainOpts {
    uint32_t channels = 1; bool dc_block = true; float agc = 0.f; bool fm = false; bool am = false; bool wbfm = false; uint32_t decim = 4; float deemph_fc = 0.025f; float kf = 0.3f; bool mix = false;
    uint32_t max_frames = 4096; uint32_t flags = CSDR_FLAG_QUIET;
    // channel shard of a multi-GPU run (one process per GPU): this process owns the channels rank, rank + world, ... (interleaved
    // ownership); with mix the partial sums meet in ONE all-reduce per chunk (csdr_chain_process_mix over `comm`)
    uint32_t world = 1, rank = 0; csdr_comm *comm = nullptr; int device = -1;       // device: HIP ordinal of this process's GPU (-1 = current)
    uint32_t owned() const { return world > 1 ? channels / world : channels; }
    uint32_t channel_of(uint32_t row) const { return world > 1 ? rank + world * row : row; }       // 0-based channel of output row `row`
};

// csdr_comm bootstrap through a file (RCCL's unique id is 128 opaque bytes that rank 0 makes and every rank needs): rank 0 writes
// `path` atomically, the others wait for it.  device -1 = current.
// The file is  "CSDRID01" | uint64 nonce | id : a waiting rank accepts it only when the nonce is its own run's, so the id a crashed
// earlier run left behind under the same path is never taken for this run's (ranks with different ids would block in
// ncclCommInitRank for ever).  nonce 0 = the launcher's pid (getppid(): the ranks of one run are children of one launcher); pass an
// explicit one (soapy_sdr_file --id-nonce N, a launcher pid or a timestamp) when the ranks do not share a parent.  Rank 0 removes
// the file once csdr_comm_create has returned -- that call is collective, so every rank has read the id by then.
inline uint64_t commRunNonce(uint64_t nonce) { return nonce ? nonce : (uint64_t)getppid(); }
inline bool commReadIdFile(const std::string &path, uint64_t nonce, unsigned char *id)
{
    unsigned char rec[16 + CSDR_COMM_ID_BYTES];
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = std::fread(rec, 1, sizeof rec, f);
    std::fclose(f);
    uint64_t have = 0;
    if (got != sizeof rec || std::memcmp(rec, "CSDRID01", 8) != 0) return false;
    std::memcpy(&have, rec + 8, 8);
    if (have != nonce) return false;                       // another run's id (stale file): keep waiting for ours
    std::memcpy(id, rec + 16, CSDR_COMM_ID_BYTES);
    return true;
}
inline void commWriteIdFile(const std::string &path, uint64_t nonce, const unsigned char *id)
{
    unsigned char rec[16 + CSDR_COMM_ID_BYTES];
    std::memcpy(rec, "CSDRID01", 8);
    std::memcpy(rec + 8, &nonce, 8);
    std::memcpy(rec + 16, id, CSDR_COMM_ID_BYTES);
    std::remove(path.c_str());                             // a stale id must not be readable while ours is being written
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    const bool ok = f && std::fwrite(rec, 1, sizeof rec, f) == sizeof rec;
    if (f) std::fclose(f);
    if (!ok) throw std::runtime_error("cannot write " + tmp);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
}
inline csdr_comm *commFromIdFile(const std::string &path, int rank, int world, int device = -1, uint64_t nonce = 0)
{
    unsigned char id[CSDR_COMM_ID_BYTES];
    nonce = commRunNonce(nonce);
    if (rank == 0) {
        check(csdr_comm_unique_id(id));
        commWriteIdFile(path, nonce, id);
    } else {
        for (int tries = 0; !commReadIdFile(path, nonce, id); tries++) {
            if (tries > 6000) throw std::runtime_error("no communicator id of this run (nonce " + std::to_string(nonce) + ") in " + path + " after 60 s");
            struct timespec ts = {0, 10 * 1000 * 1000};
            nanosleep(&ts, nullptr);
        }
    }
    csdr_comm *c = nullptr;
    const int rc = csdr_comm_create(rank, world, id, device, &c);
    if (rank == 0) std::remove(path.c_str());              // every rank has joined (or the create failed): the id is spent
    check(rc);
    return c;
}

template <class Out> Pipe<Array<cf32>, std::vector<Array<Out>>> fusedChain(const ChainOpts &o)
{
    Pipe<Array<cf32>, std::vector<Array<Out>>> p;
    p.start = [o]() {
        csdr_chain_cfg cfg;
        csdr_chain_cfg_default(&cfg, o.channels);
        cfg.channels = o.channels; cfg.dc_block = o.dc_block; cfg.agc_threshold_db = o.agc;
        cfg.demod = o.fm ? CSDR_DEMOD_FM : (o.am ? CSDR_DEMOD_AM : (o.wbfm ? CSDR_DEMOD_WBFM : CSDR_DEMOD_NONE)); cfg.wbfm_decim = o.decim; cfg.deemph_fc = o.deemph_fc; cfg.kf = o.kf; cfg.mix = o.mix; cfg.max_frames = o.max_frames; cfg.flags = o.flags;
        cfg.device = o.device;
        if (o.world > 1) {
            if (o.channels % o.world || o.rank >= o.world) throw std::runtime_error("channel shards: world must divide the channel count");
            cfg.chan_first = o.rank; cfg.chan_stride = o.world;
        }
        csdr_chain *h = nullptr;
        check(csdr_chain_create(&cfg, &h));
        return std::shared_ptr<void>(h, [](void *q) { csdr_chain_destroy(static_cast<csdr_chain *>(q)); });
    };
    p.process = [o](void *r, const Array<cf32> &a) {
        auto *h = static_cast<csdr_chain *>(r);
        const uint32_t M = o.channels;
        if (a.empty()) return std::vector<Array<Out>>{Array<Out>{}};          // nx = 0 -> [empty] (Liquid.chs:856-862)
        uint32_t usable = (uint32_t)(a.size() / M * M), nf = usable / M;
        const bool mixed = o.mix && M > 1;
        const uint32_t no = o.wbfm ? nf / o.decim : nf;                        // DeWBFM: nf div decim samples per channel
        if (o.wbfm) { nf = no * o.decim; usable = nf * M; }                    // firDecimator drops the leftover (Liquid.chs:495-497)
        const uint32_t C = o.owned();                                          // rows this process produces
        Array<Out> flat((size_t)(mixed ? no : (size_t)C * no));
        uint32_t n_out = 0;
        if (usable) {
            // --mix across ranks: local left fold over the owned channels + one all-reduce (Trans.hs:119-122 over the node)
            if (mixed && o.comm) check(csdr_chain_process_mix(h, o.comm, reinterpret_cast<const float *>(a.data()), usable, flat.data(), &n_out));
            else check(csdr_chain_process(h, reinterpret_cast<const float *>(a.data()), usable, flat.data(), &n_out));
        }
        std::vector<Array<Out>> outs;
        if (mixed || M == 1) { outs.push_back(std::move(flat)); return outs; }
        for (uint32_t k = 0; k < C; k++) outs.emplace_back(flat.begin() + (size_t)k * no, flat.begin() + (size_t)(k + 1) * no);
        return outs;
    };
    p.done = [](void *) {};
    return p;
}

}  // namespace csdrhost
