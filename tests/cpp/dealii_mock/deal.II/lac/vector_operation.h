#pragma once
namespace dealii
{
  struct VectorOperation { enum values { unknown, insert, add, min, max }; };
}
