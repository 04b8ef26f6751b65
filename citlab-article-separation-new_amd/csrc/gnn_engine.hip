// GNN relation predictor engine -- entry points (kernels follow in the next milestone).
#include "asep_common.h"
#include "gnn_kernels.h"
using namespace asep;
struct asep_gnn { asep_gnn_cfg cfg; };
extern "C" {
asep_gnn* asep_gnn_load(const void*, size_t, const asep_gnn_cfg*) { set_error("GNN engine not built yet"); return nullptr; }
void asep_gnn_free(asep_gnn* g) { delete g; }
int asep_gnn_correct_edges(asep_gnn*, int, int, const int32_t*, const float*, int32_t*, float*) { set_error("GNN engine not built yet"); return ASEP_ERR_UNSUPPORTED; }
int asep_gnn_forward(asep_gnn*, int, int, const int32_t*, const float*, const float*, int, const int32_t*, float*) { set_error("GNN engine not built yet"); return ASEP_ERR_UNSUPPORTED; }
int asep_gnn_forward_dev(asep_gnn*, int, int, const int32_t*, const float*, const float*, int, const int32_t*, float*, void*) { set_error("GNN engine not built yet"); return ASEP_ERR_UNSUPPORTED; }
int asep_gnn_get_hidden(asep_gnn*, float*, size_t) { return ASEP_ERR_UNSUPPORTED; }
double asep_gnn_flops(const asep_gnn*, int, int, int) { return 0.0; }
}
