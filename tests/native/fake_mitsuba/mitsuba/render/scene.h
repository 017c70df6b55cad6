#include <mitsuba/fake_mitsuba.h>
