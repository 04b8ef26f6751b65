// GNN device kernels (placeholder until the message-passing kernels land).
#pragma once
