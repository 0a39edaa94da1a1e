// wp_stream_unit.hip - streaming weight pool of the bf16 large-graph path (wp_stream.h)
#include "wp_stream.h"
