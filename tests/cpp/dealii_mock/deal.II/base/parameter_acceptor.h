#pragma once
#include <deal.II/base/exceptions.h>
#include <deal.II/base/subscriptor.h>
#include <algorithm>
#include <functional>
#include <map>
#include <set>
#include <type_traits>
#include <memory>
#include <string>
#include <tuple>
#include <vector>
namespace dealii
{
  namespace Patterns
  {
    class PatternBase
    {
    public:
      virtual ~PatternBase() = default;
      virtual bool match(const std::string &) const { return true; }
      enum OutputStyle { Machine, Text, LaTeX };
      virtual std::string description(const OutputStyle = Machine) const { return {}; }
      virtual std::unique_ptr<PatternBase> clone() const { return nullptr; }
    };
    class Selection : public PatternBase
    {
    public:
      explicit Selection(const std::string &) {}
    };
    class Anything : public PatternBase {};
    class Bool : public PatternBase {};
    class Double : public PatternBase { public: Double(double = 0., double = 0.) {} };
    class Integer : public PatternBase { public: Integer(int = 0, int = 0) {} };
    class List : public PatternBase { public: List(const PatternBase &, unsigned int = 0, unsigned int = 0, const std::string & = ",") {} };
    namespace Tools
    {
      template <class T, class Enable = void>
      struct Convert {
        static std::unique_ptr<Patterns::PatternBase> to_pattern() { return std::make_unique<Patterns::Anything>(); }
        static std::string to_string(const T &, const Patterns::PatternBase & = *Convert<T>::to_pattern()) { return {}; }
        /* numbers, bool and strings are parsed (what tests/cpp/time_integrator_run.cc sets); ryujin's enums bring
         * their own specialisation (source/patterns_conversion.h); anything else keeps its default */
        static T to_value(const std::string &s, const Patterns::PatternBase & = *Convert<T>::to_pattern())
        {
          if constexpr (std::is_same<T, bool>::value)
            return s == "true" || s == "1" || s == "yes";
          else if constexpr (std::is_arithmetic<T>::value)
            return static_cast<T>(std::stod(s));
          else if constexpr (std::is_same<T, std::string>::value)
            return s;
          else
            return T();
        }
      };
      struct ExcNoMatch : ExceptionBase { ExcNoMatch(const std::string &, const std::string &) {} };
    }
    using Tools::ExcNoMatch;
  }
  using Patterns::Tools::ExcNoMatch;

  class ParameterHandler
  {
  public:
    void enter_subsection(const std::string &);
    void leave_subsection();
    void declare_entry(const std::string &, const std::string &, const Patterns::PatternBase & = Patterns::Anything(), const std::string & = "");
    std::string get(const std::string &) const;
    double get_double(const std::string &) const;
    long get_integer(const std::string &) const;
    bool get_bool(const std::string &) const;
    template <class T> void add_parameter(const std::string &, T &, const std::string & = "", const Patterns::PatternBase & = *Patterns::Tools::Convert<T>::to_pattern());
    void add_action(const std::string &, const std::function<void(const std::string &)> &);
    enum OutputStyle { Text = 1, LaTeX = 2, Description = 4, XML = 8, JSON = 16, PRM = 32, Short = 64, KeepDeclarationOrder = 128, ShortPRM = 192 | 32 };
    std::ostream &print_parameters(std::ostream &, const unsigned int) const;
    void print_parameters(const std::string &, const unsigned int) const;
    void log_parameters(class LogStream &, const unsigned int = 0);
    /* mock run time: what a .prm file would hold, keyed by (section, entry); applied by ParameterAcceptor::initialize() */
    void set(const std::string &section, const std::string &entry, const std::string &value) { values[section + "\n" + entry] = value; }
    const std::string *find(const std::string &section, const std::string &entry) const
    {
      const auto it = values.find(section + "\n" + entry);
      return it == values.end() ? nullptr : &it->second;
    }
    std::map<std::string, std::string> values;
    std::set<std::string> consumed;
  };

  class ParameterAcceptor : public Subscriptor
  {
  public:
    explicit ParameterAcceptor(const std::string &section_name = "") : section(section_name) { registry().push_back(this); }
    virtual ~ParameterAcceptor()
    {
      auto &r = registry();
      r.erase(std::remove(r.begin(), r.end(), this), r.end());
    }
    /* apply what ParameterHandler::set() recorded to every registered object, then fire parse_parameters_call_back
     * (what deal.II does after reading the .prm file) */
    static void initialize(const std::string & = "", const std::string & = "",
                           const ParameterHandler::OutputStyle = ParameterHandler::Short, ParameterHandler &handler = ParameterAcceptor::prm,
                           const ParameterHandler::OutputStyle = ParameterHandler::Short)
    {
      for (ParameterAcceptor *a : registry()) {
        for (auto &setter : a->setters_)
          setter(handler);
        a->parse_parameters(handler);
        a->parse_parameters_call_back();
      }
      for (const auto &it : handler.values)
        if (!handler.consumed.count(it.first))
          throw ExcMessage("mock ParameterAcceptor: no such entry: " + it.first);
    }
    virtual void declare_parameters(ParameterHandler &) {}
    virtual void parse_parameters(ParameterHandler &) {}
    struct Signal {
      template <typename F> void connect(F &&f) { slots.emplace_back(std::forward<F>(f)); }
      void operator()() const { for (const auto &f : slots) f(); }
      std::vector<std::function<void()>> slots;
    };
    Signal declare_parameters_call_back;
    Signal parse_parameters_call_back;
    std::string get_section_name() const { return section; }
    std::vector<std::string> get_section_path() const { return {}; }
    template <class ParameterType>
    void add_parameter(const std::string &entry, ParameterType &parameter, const std::string & = "", ParameterHandler & = prm,
                       const Patterns::PatternBase & = *Patterns::Tools::Convert<ParameterType>::to_pattern())
    {
      if constexpr (!std::is_const<ParameterType>::value) {
        ParameterType *target = &parameter;
        const std::string sec = section;
        setters_.push_back([target, sec, entry](ParameterHandler &handler) {
          if (const std::string *value = handler.find(sec, entry)) {
            *target = Patterns::Tools::Convert<ParameterType>::to_value(*value);
            handler.consumed.insert(sec + "\n" + entry);
          }
        });
      }
    }
    void enter_subsection(const std::string &) {}
    void leave_subsection() {}
    void enter_my_subsection(ParameterHandler & = prm) {}
    void leave_my_subsection(ParameterHandler & = prm) {}
    inline static ParameterHandler prm;
  protected:
    const std::string section;
  private:
    static std::vector<ParameterAcceptor *> &registry()
    {
      static std::vector<ParameterAcceptor *> r;
      return r;
    }
    std::vector<std::function<void(ParameterHandler &)>> setters_;
  };
}
