package ipx

import "runtime"

// runtimePinner is runtime.Pinner (Go 1.21+): it keeps Go memory where it is while C reads it.
type runtimePinner = runtime.Pinner
