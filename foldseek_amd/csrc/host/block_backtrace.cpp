// placeholder until the block-aligner restatement lands (next commit)
#include "hostlib.h"
namespace fsh {
void blockBacktrace(const Matrix &, const Matrix &, const uint8_t *, const uint8_t *, const int8_t *, const int8_t *, int,
                    const uint8_t *, const uint8_t *, int, int, int, int, int, int, BlockAlnOut &out) { out.ok = false; }
}
