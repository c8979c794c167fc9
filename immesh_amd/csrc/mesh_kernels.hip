// mesher kernels (placeholder TU until implemented)
