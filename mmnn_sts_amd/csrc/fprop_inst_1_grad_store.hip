// transition conv data gradient: explicit instantiation of the tile dispatch (see fprop_dispatch.hpp)
#include "fprop_dispatch.hpp"

namespace mmnn {
template int dispatch<1, PRO_GRAD, EPI_STORE>(const FpropArgs&, hipStream_t);
}  // namespace mmnn
