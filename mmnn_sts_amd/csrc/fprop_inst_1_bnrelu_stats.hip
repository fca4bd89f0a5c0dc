// dense-layer conv1 forward: explicit instantiation of the tile dispatch (see fprop_dispatch.hpp)
#include "fprop_dispatch.hpp"

namespace mmnn {
template int dispatch<1, PRO_BNRELU, EPI_STORE_STATS>(const FpropArgs&, hipStream_t);
}  // namespace mmnn
