// Stand-in for SDR++ core's dsp/processor.h (+ types.h, block.h, stream.h, buffer/buffer.h, taps/tap.h, math helpers, the two
// VOLK dot products).  Own restatement from SURVEY.md Appendix A -- see README.md.  Single-threaded: streams never block.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cassert>
#include <mutex>
#include <vector>

#define FL_M_PI 3.1415926535f
#define STREAM_BUFFER_SIZE 1000000

// How the tap generators spell pi.  Default: the double constant (SURVEY.md Appendix A as first written).  With
// -DREFSHIM_FLOAT_PI: the FLOAT macro FL_M_PI inside the double expressions -- how upstream SDR++ is recalled to write
// root_raised_cosine.h, windowed_sinc.h, the cosine windows and hzToRads; which of the two upstream really
// does cannot be checked in this image, so both variants exist and tests/test_sdrpp_tables.py measures what hangs on it (most taps
// change their bit pattern, by <= 1e-4 relative; no bit after lock does).  A real SDR++ build settles it by construction: the
// block designs with the INSTALLED headers (host/sdrpp_tables.h).
#ifdef REFSHIM_FLOAT_PI
#define REFSHIM_PI FL_M_PI
#else
#define REFSHIM_PI 3.14159265358979323846
#endif

namespace dsp {
    struct complex_t {
        float re, im;
        complex_t operator*(const float b) const { return complex_t{ re * b, im * b }; }
        complex_t operator*(const complex_t& b) const { return complex_t{ re * b.re - im * b.im, im * b.re + re * b.im }; }
        complex_t operator+(const complex_t& b) const { return complex_t{ re + b.re, im + b.im }; }
        complex_t operator-(const complex_t& b) const { return complex_t{ re - b.re, im - b.im }; }
        float phase() const { return atan2f(im, re); }
        float amplitude() const { return sqrtf(re * re + im * im); }
        float fastAmplitude() const {
            const float r = fabsf(re), i = fabsf(im);
            return r > i ? r + 0.4f * i : i + 0.4f * r;
        }
    };

    namespace buffer {
        template <class T> T* alloc(int count) { return (T*)calloc((size_t)count, sizeof(T)); }
        template <class T> void clear(T* p, int count, int offset = 0) { memset(&p[offset], 0, (size_t)count * sizeof(T)); }
        template <class T> void free(T* p) { ::free((void*)p); }
    }

    template <class T> struct tap {
        T* taps = nullptr;
        int size = 0;
    };
    namespace taps {
        template <class T> tap<T> alloc(int count) {
            tap<T> t;
            t.taps = buffer::alloc<T>(count);
            t.size = count;
            return t;
        }
        template <class T> void free(tap<T>& t) {
            if (t.taps) { buffer::free(t.taps); }
            t.taps = nullptr;
            t.size = 0;
        }
    }

    namespace math {
        inline double sinc(double x) { return x == 0.0 ? 1.0 : sin(x) / x; }
        inline complex_t phasor(float x) { return complex_t{ cosf(x), sinf(x) }; }
        template <class T> inline T step(T x) { return x > (T)0 ? (T)1 : (T)-1; }
        inline double hzToRads(double f, double fs) { return 2.0 * REFSHIM_PI * (f / fs); }
    }

    // stream: never blocks here; read() reports "stopped" because nothing drives it.
    template <class T> class stream {
    public:
        stream() { writeBuf = buffer::alloc<T>(STREAM_BUFFER_SIZE); readBuf = buffer::alloc<T>(STREAM_BUFFER_SIZE); }
        ~stream() { free(); }
        void free() {
            if (writeBuf) { buffer::free(writeBuf); }
            if (readBuf) { buffer::free(readBuf); }
            writeBuf = readBuf = nullptr;
        }
        int read() { return -1; }
        void flush() {}
        bool swap(int) { std::swap(writeBuf, readBuf); return true; }
        T* writeBuf = nullptr;
        T* readBuf = nullptr;
    };

    class block {
    public:
        virtual ~block() {}
        virtual int run() = 0;
        void start() {}
        void stop() {}
        void tempStop() {}
        void tempStart() {}
        bool _block_init = false;
        std::recursive_mutex ctrlMtx;
    };

    template <class I, class O> class Processor : public block {
    public:
        Processor() {}
        virtual void init(stream<I>* in) { _in = in; _block_init = true; }
        stream<O> out;
    protected:
        stream<I>* _in = nullptr;
    };
}

// VOLK stand-ins: plain ascending-index scalar accumulation (VOLK's real order depends on the dispatched SIMD kernel).
typedef dsp::complex_t lv_32fc_t;
inline void volk_32fc_32f_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* in, const float* taps, unsigned int n) {
    float re = 0.0f, im = 0.0f;
    for (unsigned int k = 0; k < n; k++) { re += in[k].re * taps[k]; im += in[k].im * taps[k]; }
    result->re = re;
    result->im = im;
}
inline void volk_32fc_x2_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* in, const lv_32fc_t* taps, unsigned int n) {
    float re = 0.0f, im = 0.0f;
    for (unsigned int k = 0; k < n; k++) {
        re += in[k].re * taps[k].re - in[k].im * taps[k].im;
        im += in[k].im * taps[k].re + in[k].re * taps[k].im;
    }
    result->re = re;
    result->im = im;
}
