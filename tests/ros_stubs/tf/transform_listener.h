#pragma once
namespace tf { struct TransformListener {}; }
