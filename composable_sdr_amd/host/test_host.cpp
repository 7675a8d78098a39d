// CPU-only checks of the C++ combinators against Trans.hs / Types.hs semantics (no GPU calls).
#include <cassert>
#include <iostream>

#include "csdr_host.hpp"

using namespace csdrhost;

template <class T> struct Collect : Fold<Array<T>> {
    std::vector<Array<T>> items; bool finished = false;
    void step(const Array<T> &a) override { items.push_back(a); }
    void done() override { finished = true; }
};

static Array<float> iota(int a, int b) { Array<float> v; for (int i = a; i < b; i++) v.push_back((float)i); return v; }

int main()
{
    {   // compact (Trans.hs:58-84)
        auto sink = std::make_shared<Collect<float>>();
        auto f = compact<float>(5, sink);
        f->step(iota(0, 3)); f->step(iota(3, 6)); f->step(iota(6, 7)); f->step(iota(7, 13)); f->done();
        assert(sink->items.size() == 3 && sink->items[0].size() == 5 && sink->items[1].size() == 5 && sink->items[2].size() == 3);
        assert(sink->items[1][0] == 5.f && sink->items[2][2] == 12.f && sink->finished);
        auto s2 = std::make_shared<Collect<float>>();
        auto g = compact<float>(4, s2);
        g->step(iota(0, 4)); g->done();
        assert(s2->items.size() == 2 && s2->items[1].empty());          // empty remainder still pushed
        auto s3 = std::make_shared<Collect<float>>();
        auto h = compact<float>(4, s3);
        h->step(iota(0, 11)); h->done();
        assert(s3->items.size() == 2 && s3->items[0].size() == 4 && s3->items[1].size() == 7);
    }
    {   // takeNArr (Trans.hs:33-56)
        TakeN t(6);
        auto a = iota(0, 4), b = iota(4, 8), c = iota(8, 12);
        assert(t.feed(a) && a.size() == 4);
        assert(t.feed(b) && b.size() == 2);
        assert(!t.feed(c));
    }
    {   // communicator bootstrap file: another run's id (stale file, other nonce), a short file and a foreign file are not accepted
        const std::string path = "/tmp/csdr_test_host_" + std::to_string((long)getpid()) + ".id";
        unsigned char id[CSDR_COMM_ID_BYTES], got[CSDR_COMM_ID_BYTES];
        for (size_t i = 0; i < sizeof id; i++) id[i] = (unsigned char)(i * 7 + 1);
        commWriteIdFile(path, 41, id);
        assert(!commReadIdFile(path, 42, got));                       // the previous run's
        assert(commReadIdFile(path, 41, got) && std::memcmp(id, got, sizeof id) == 0);
        commWriteIdFile(path, 42, id);                                // rank 0 of the new run replaces it
        assert(commReadIdFile(path, 42, got) && !commReadIdFile(path, 41, got));
        { FILE *f = std::fopen(path.c_str(), "wb"); std::fwrite(id, 1, sizeof id, f); std::fclose(f); }     // round 5's format: 128 bare bytes
        assert(!commReadIdFile(path, 42, got));
        std::remove(path.c_str());
        assert(!commReadIdFile(path, 42, got));
        assert(commRunNonce(0) == (uint64_t)getppid() && commRunNonce(9) == 9);
    }
    {   // mix = left fold (Trans.hs:119-122)
        std::vector<Array<float>> ch = {{1e8f, 1.f}, {-1e8f, 1.f}, {1.f, 1.f}};
        auto m = mix(ch);
        assert(m[0] == 1.f && m[1] == 3.f);
    }
    {   // compose / addPipe life-cycle (Types.hs:93-131)
        std::string log;
        Pipe<Array<float>, Array<float>> a, b;
        a.start = [&]() { log += "sA"; return std::shared_ptr<void>(); };
        a.process = [](void *, const Array<float> &x) { auto y = x; for (auto &v : y) v += 1; return y; };
        a.done = [&](void *) { log += "dA"; };
        b.start = [&]() { log += "sB"; return std::shared_ptr<void>(); };
        b.process = [](void *, const Array<float> &x) { auto y = x; for (auto &v : y) v *= 2; return y; };
        b.done = [&](void *) { log += "dB"; };
        auto c = compose<Array<float>, Array<float>, Array<float>>(a, b);        // a . b : b first
        auto r = c.start();
        auto y = c.process(r.get(), Array<float>{1.f, 2.f});
        c.done(r.get());
        assert(y[0] == 3.f && y[1] == 5.f && log == "sAsBdBdA");
        auto sink = std::make_shared<Collect<float>>();
        auto f = addPipe<Array<float>, Array<float>>(b, sink);
        f->step(Array<float>{1.f}); f->done();
        assert(sink->items[0][0] == 2.f && sink->finished);
    }
    {   // unPipe (Types.hs:109-115): resource created at unPipe, cleanup = done; idPipe is the Category id
        std::string log;
        Pipe<Array<float>, Array<float>> a;
        a.start = [&]() { log += "s"; return std::shared_ptr<void>(); };
        a.process = [](void *, const Array<float> &x) { auto y = x; for (auto &v : y) v += 1; return y; };
        a.done = [&](void *) { log += "d"; };
        auto u = unPipe(compose<Array<float>, Array<float>, Array<float>>(a, idPipe<Array<float>>()));
        assert(log == "s");
        auto y = u.process(Array<float>{1.f});
        auto z = u.process(y);
        assert(z[0] == 3.f && log == "s");
        u.cleanup();
        assert(log == "sd");
    }
    {   // distribute_ (Trans.hs:106-117)
        auto s0 = std::make_shared<Collect<float>>(), s1 = std::make_shared<Collect<float>>();
        Distribute<float> d({s0, s1});
        d.step({iota(0, 2), iota(10, 12)});
        d.step({iota(2, 3)});                                          // one element: sink 1 only
        d.done();
        assert(s0->items.size() == 2 && s1->items.size() == 1 && s0->finished && s1->finished);
    }
    std::cout << "host combinators ok\n";
    return 0;
}
