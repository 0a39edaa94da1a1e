// translation unit of the streaming weight-gradient kernel (kept apart from engine.hip: rebuilds in seconds)
#include "wgrad_stream.h"
