// render.hip -- wavefront path tracing pipeline (placeholder until the pipeline lands).
#include "scene.h"

extern "C" int ptx_render_tiles(ptx_scene_t, const PtxCamera*, const PtxTile*, uint32_t, uint64_t, int64_t*, uint32_t*, uint32_t) {
	return ptx::fail(PTX_ERR_UNSUPPORTED, "ptx_render_tiles: not built yet");
}
