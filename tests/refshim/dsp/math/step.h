// Stand-in: math::step lives in dsp/processor.h here.
#pragma once
#include <dsp/processor.h>
