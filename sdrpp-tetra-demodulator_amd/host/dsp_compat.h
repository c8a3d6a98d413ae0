// dsp_compat.h -- the few SDR++ core types the demodulator block touches, for builds OUTSIDE SDR++.
//
// Inside an SDR++ module build define TETRA_WITH_SDRPP and the real <dsp/processor.h> is used instead
// (the block then plugs into the host's streams exactly like the reference's src/dsp/pi4dqpsk.h does).
// These stand-ins are this project's own minimal code (single-producer/single-consumer swap stream, a
// worker-thread block); they exist so that the drop-in class and its tests compile without SDR++.
#pragma once
#ifdef TETRA_WITH_SDRPP
#include <dsp/processor.h>
#else
#include <atomic>
#include <cassert>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define STREAM_BUFFER_SIZE 1000000

namespace dsp {
struct complex_t {
    float re, im;
};

// Double-buffered blocking stream: writer fills writeBuf then swap(n); reader read()s n items from readBuf
// and flush()es.  read() < 0 / swap() == false once stopped.
template <class T> class stream {
public:
    stream() : wb(STREAM_BUFFER_SIZE), rb(STREAM_BUFFER_SIZE) { writeBuf = wb.data(); readBuf = rb.data(); }
    T* writeBuf;
    T* readBuf;
    bool swap(int n) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !ready || stopW; });
        if (stopW) return false;
        std::swap(wb, rb);
        writeBuf = wb.data();
        readBuf = rb.data();
        count = n;
        ready = true;
        cv.notify_all();
        return true;
    }
    int read() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return ready || stopR; });
        return stopR ? -1 : count;
    }
    void flush() {
        std::lock_guard<std::mutex> l(m);
        ready = false;
        cv.notify_all();
    }
    void stopWriter() { std::lock_guard<std::mutex> l(m); stopW = true; cv.notify_all(); }
    void stopReader() { std::lock_guard<std::mutex> l(m); stopR = true; cv.notify_all(); }
    void clearStops() { std::lock_guard<std::mutex> l(m); stopW = stopR = false; }
    void free() {}

private:
    std::vector<T> wb, rb;
    std::mutex m;
    std::condition_variable cv;
    int count = 0;
    bool ready = false, stopW = false, stopR = false;
};

class block {
public:
    virtual ~block() {}
    virtual int run() = 0;
    virtual void start() {
        if (running) return;
        running = true;
        doStart();
    }
    virtual void stop() {
        if (!running) return;
        doStop();
        running = false;
    }
    void tempStop() { if (tempStopDepth++ == 0 && running) { doStop(); tempStopped = true; } }
    void tempStart() { if (tempStopDepth && --tempStopDepth == 0 && tempStopped) { doStart(); tempStopped = false; } }

protected:
    virtual void doStart() { worker = std::thread([this] { while (run() >= 0) {} }); }
    virtual void doStop() = 0;
    void joinWorker() { if (worker.joinable()) worker.join(); }
    bool _block_init = false;
    std::recursive_mutex ctrlMtx;
    bool running = false, tempStopped = false;
    int tempStopDepth = 0;
    std::thread worker;
};

template <class I, class O> class Processor : public block {
public:
    virtual void init(stream<I>* in) { _in = in; _block_init = true; }
    stream<O> out;

protected:
    void doStop() override {
        if (_in) _in->stopReader();
        out.stopWriter();
        joinWorker();
        if (_in) _in->clearStops();
        out.clearStops();
    }
    stream<I>* _in = nullptr;
};
}  // namespace dsp
#endif
