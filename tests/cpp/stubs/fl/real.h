// STAND-IN (tests/cpp/stubs/README.md): fl::Real is double unless fl is built otherwise.
#pragma once
namespace fl { typedef double Real; }
