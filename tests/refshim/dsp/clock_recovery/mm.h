// Stand-in: the reference includes dsp/clock_recovery/mm.h but uses nothing from it on this path.
#pragma once
#include <dsp/processor.h>
