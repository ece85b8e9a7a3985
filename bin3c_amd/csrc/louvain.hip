// placeholder until the Louvain kernels land
#include "common.hpp"
extern "C" int b3c_louvain(b3c_ctx *ctx, const b3c_mat *, double, int, int, int32_t *, b3c_louvain_info *)
{
    return ctx ? ctx->fail(B3C_E_ARG, "b3c_louvain: not built yet") : B3C_E_ARG;
}
extern "C" int b3c_graph_quantise(b3c_ctx *ctx, const b3c_mat *, double, int64_t *, int32_t *)
{
    return ctx ? ctx->fail(B3C_E_ARG, "b3c_graph_quantise: not built yet") : B3C_E_ARG;
}
